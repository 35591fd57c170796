"""Full MLP forward at the C2 point count with and without the feature output (the with-gradient render writes it for the backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from test_gpu_backward import NET_KW, SDF_NPZ
from oi_amd import ops
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
col_sd = load_golden("weights_color")
sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda(); col_net = ColorNetwork(**NET_KW); col_net.load_state_dict(col_sd); col_net = col_net.cuda()
for mode in ("f16x3", "bf16"):
    pack = FieldPack(sdf_net, col_net, mode)
    B, n = 1, 524288
    pts = (torch.rand(B * n, 3, device="cuda") * 2 - 1)
    with torch.no_grad():
        _, gamma, beta = pack.film(w=torch.randn(B, 64, device="cuda"))
        for feat in (False, True, False, True):
            f = lambda: ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig, True, True, feat, None)
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): f()
            torch.cuda.synchronize()
            print(f"{mode} want_feat={feat}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
