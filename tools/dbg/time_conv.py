"""Kernel time of the small-batch convolution kernels at the shapes of a 64^2 discriminator (n_feat 512) with two images:
forward, data gradient, weight gradient per layer (HIP events over 200 back-to-back launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
from oi_amd import ops
B = int(os.environ.get("OI_DBG_B", 2))
chans, res = [3, 64, 128, 256, 512], [64, 32, 16, 8, 4]
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for l in range(4):
    Cin, Cout, H = chans[l], chans[l + 1], res[l]
    x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
    g = torch.randn(B, Cout, H // 2, H // 2, device="cuda")
    gf = 2 * B * Cout * Cin * 16 * (H // 2) ** 2 / 1e9
    print(f"layer {l}: {Cin:3d}->{Cout:3d} @{H:2d}  {gf:5.2f} GFLOP  weights {Cout * Cin * 64 / 1e6:5.2f} MB   fwd {t(lambda: ops.conv4x4_fwd(x, w, None, 2, 1, 1.0)):6.1f} us   "
          f"dgrad {t(lambda: ops.conv4x4_dgrad(g, w, H, H, 2, 1)):6.1f} us   wgrad {t(lambda: ops.conv4x4_wgrad(g, x, 2, 1)):6.1f} us   "
          f"bwd {t(lambda: ops.conv4x4_bwd(g, w, x, 2, 1)):6.1f} us")
