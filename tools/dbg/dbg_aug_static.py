import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT + "/object-intrinsics_amd")
import numpy as np, torch
from oi_amd.augment import AugmentPipe
aug = AugmentPipe(xint=1, scale=1).cuda()
H = W = 64
np.random.seed(0); torch.manual_seed(0)
worst = (0, 0)
for trial in range(300):
    x = torch.rand(1, 3, H, W, device="cuda")
    r = torch.randn(1, 3, H, W, device="cuda")
    G = aug.sample_G_inv(x)
    outs = []
    for static in (False, True):
        m = aug.static_margins(H, W) if static else aug.margins_for(G, H, W)
        th = torch.from_numpy(aug.theta_for(G, m, H, W)).cuda()
        xi = x.clone().requires_grad_()
        y = aug.apply_theta(xi, th, m)
        (gx,) = torch.autograd.grad((y * r).sum(), xi)
        outs.append((y.detach(), gx, m))
    dy = float((outs[0][0] - outs[1][0]).abs().max()); dg = float((outs[0][1] - outs[1][1]).abs().max())
    fin = bool(torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all())
    if dy > 1e-3 or dg > 1e-3 or not fin:
        print(trial, "margins", outs[0][2], "dy", dy, "dg", dg, "finite", fin, "G", G[0].round(3).tolist())
    worst = (max(worst[0], dy), max(worst[1], dg))
print("worst", worst)
