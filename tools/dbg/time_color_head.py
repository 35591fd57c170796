"""Stand-alone albedo head (ColorNetwork.forward on caller-supplied features, csrc/color_head.hip) at the C2 point count:
forward and backward time from HIP events, against the fp32-MFMA peak (157.3 TFLOP/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
from oi_amd.fields import ColorNetwork
n = int(os.environ.get("OI_DBG_N", 4096 * 128))
col = ColorNetwork(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64).cuda()
feat = (torch.rand(n, 128, device="cuda") * 2 - 1).requires_grad_(True)
nrm = torch.randn(n, 3, device="cuda").requires_grad_(True)
w = torch.randn(1, 64, device="cuda")
pts = torch.zeros(n, 3, device="cuda")
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
with torch.no_grad():
    f_ms = t(lambda: col(pts, nrm, None, feat, None, w))
def fb():
    rgb = col(pts, nrm, None, feat, None, w)
    torch.autograd.grad(rgb.sum(), [feat, nrm] + list(col.parameters()))
fb_ms = t(fb, 10)
F = 2 * 131 * 128 + 2 * 128 * 3
print(f"points {n}: forward {f_ms:.3f} ms = {n * F / f_ms / 1e9:.1f} TFLOP/s ({n * F / f_ms / 1e9 / 157.3:.2f} of the fp32-MFMA peak); "
      f"forward + backward {fb_ms:.3f} ms (backward: 3 products + the 132-column point GEMM = {n * (3 * 2 * 131 * 128 + 2 * 128 * 135 + 2 * 128 * 3) / (fb_ms - f_ms) / 1e9:.1f} TFLOP/s)")
