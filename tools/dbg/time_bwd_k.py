"""Kernel times of one MLP backward at C2 size (sweep / wgrad) from HIP events around oi_sdf_mlp_bwd via the autograd path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from test_gpu_backward import NET_KW, SDF_NPZ
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
from oi_amd.autograd import sdf_mlp
col_sd = load_golden("weights_color")
sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda(); col_net = ColorNetwork(**NET_KW); col_net.load_state_dict(col_sd); col_net = col_net.cuda()
pack = FieldPack(sdf_net, col_net, "f16x3")
if os.environ.get("OI_DBG_FAST"): pack.fast_trig = True
B, n = 1, int(os.environ.get('OI_DBG_N', 524288))
pts = (torch.rand(B * n, 3, device="cuda") * 2 - 1)
w = torch.randn(B, 64, device="cuda").requires_grad_(True)
ts = []
for it in range(int(os.environ.get('OI_DBG_LOOP', 12))):
    _, gamma, beta = pack.film(w=w)
    sdf, grad, rgb, _ = sdf_mlp(pack, pts, gamma, beta, B, True, True, False)
    loss = sdf.sum() + grad.sum() + rgb.sum()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[2:])
print(f"backward (sweep + wgrad + glue) ms: min {ts[0]:.3f} median {ts[len(ts)//2]:.3f}")
