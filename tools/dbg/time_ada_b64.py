"""AugmentPipe geometry at batch 64 (3 x 64 x 64): the two-launch form against the separable one-launch form (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import numpy as np, torch
import oi_amd.augment as A
import oi_amd.ops as O
aug = A.AugmentPipe(xint=1, scale=1).cuda()
for B in (64, 2):
    x = torch.rand(B, 3, 64, 64, device="cuda")
    np.random.seed(0)
    G = aug.sample_G_inv(x)
    for static in (False, True):
        m = aug.static_margins(64, 64) if static else aug.margins_for(G, 64, 64)
        th = torch.from_numpy(aug.theta_for(G, m, 64, 64)).cuda()
        for sep in (False, True):
            O.ADA_SEPARABLE = sep
            with torch.no_grad():
                for _ in range(5): y = aug.apply_theta(x, th, m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(200): y = aug.apply_theta(x, th, m)
                e1.record(); torch.cuda.synchronize()
            print(f"B={B} static={int(static)} margins={m} separable={int(sep)}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call")
