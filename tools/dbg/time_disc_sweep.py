"""ADADiscriminatorView forward at B = 1, 4, 64: GPU-side time per call (HIP events around a run of calls: includes launch gaps when
the host cannot keep up) next to the host's own time per call (no sync inside)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
from oi_amd.config import build_from_config
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
disc = build_from_config(net("src.models.discriminator.ADADiscriminatorView",
                             aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=64,
                             in_dim=3, last_bias=False, n_feat=512, out_dim=7, out_dim_latent=0, out_dim_position=6)).cuda().eval()
for rep in range(2):
    for B in (1, 4, 64):
        x = torch.rand(B, 3, 64, 64, device="cuda")
        with torch.no_grad():
            for _ in range(5): disc(x, it=0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            t0 = time.perf_counter(); e0.record()
            for _ in range(n): disc(x, it=0)
            e1.record(); th = time.perf_counter() - t0
            torch.cuda.synchronize()
        print(f"B={B}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call by events, host {th / n * 1e6:.1f} us per call")
