#!/usr/bin/env python3
"""Host-side rate of the reference's own discriminator call (needs a GPU): ADADiscriminatorView.forward, batch 1, no gradient,
shipped augmentation -- with the parameters drawn inside the library (default) and by numpy (FAST_ADA = False), plus
oi_amd.graphed.GraphedDForward.   python tools/bench_disc_eager.py [--res 64|128]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "object-intrinsics_amd"))
import torch  # noqa: E402
import oi_amd.discriminator as DM  # noqa: E402
from oi_amd.config import build_from_config  # noqa: E402
from oi_amd.graphed import GraphedDForward  # noqa: E402

R = int(sys.argv[sys.argv.index("--res") + 1]) if "--res" in sys.argv else 64
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
disc = build_from_config(net("src.models.discriminator.ADADiscriminatorView", out_dim_latent=0, out_dim_position=6,
                             aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=R, in_dim=3,
                             last_bias=False, n_feat=512, out_dim=7)).cuda().eval()
x = torch.rand(1, 3, R, R, device="cuda")


def rate(f, n=3000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


with torch.no_grad():
    for fast in (True, False, True, False):
        DM.FAST_ADA = fast
        DM._FAST_ADA.pop(disc, None)
        print(f"eager forward, FAST_ADA={fast}: {rate(lambda: disc(x, it=0)):9.0f} images/s")
    DM.FAST_ADA = True
    gd = GraphedDForward(disc)
    print(f"GraphedDForward:              {rate(lambda: gd(x)):9.0f} images/s")
