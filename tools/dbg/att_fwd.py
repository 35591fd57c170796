"""One small forward launch of the register-resident kernel (for rocprofv3 --att)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from oi_amd import ops
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
sdf = ShapeNetwork(os.path.join(ROOT, "tests", "golden", "weights_sdf.npz"), **kw).cuda()
col = ColorNetwork(**kw); col.load_state_dict(load_golden("weights_color")); col = col.cuda()
n = int(os.environ.get("OI_DBG_N", 1 << 17))
pts = (torch.rand(n, 3, device="cuda") * 2 - 1) * 0.9
pack = FieldPack(sdf, col, "f16x3")
with torch.no_grad():
    _, gamma, beta = pack.film(z=torch.randn(1, 64, device="cuda"))
    out = None
    for _ in range(3):
        out = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, True, True, False, out[-1] if out else None)
torch.cuda.synchronize()
print("ok")
