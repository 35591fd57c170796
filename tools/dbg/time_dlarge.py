"""Per-kernel time of the batch >= 16 discriminator forward (csrc/disc_large.hip) from HIP events; OI_LIB selects a variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
import oi_amd.discriminator as DM
B = int(os.environ.get("OI_DBG_B", 64))
D = DM.DCDiscriminator(in_dim=3, out_dim=7, n_feat=512, img_size=64).cuda().eval()
x = torch.rand(B, 3, 64, 64, device="cuda")
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    print(os.environ.get("OI_LIB", "default"), f"B={B} network forward {t(lambda: D(x)):7.1f} us")
