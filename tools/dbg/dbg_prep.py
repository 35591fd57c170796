import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests", ROOT + "/oracle"):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_modules import build_generator
from oi_amd import ops
gen = build_generator(16, 16, 16, 1, "f16x3").train()
bs = 2
np.random.seed(5); torch.manual_seed(5)
b2w_h, w2b_h, c2b_h, xy_h, bg_h = gen._sample_prior_host(bs, {})
z = torch.randn(bs, 64, device="cuda")
R, S = 16, 16
jit = torch.rand(bs * R * R, 1, device="cuda")
P = gen.renderer.pack.film_stacked(differentiable=False)
kinv = gen._kinv(z.device)
pre = ops.prep_render(b2w_h, w2b_h, c2b_h, xy_h, bg_h, kinv, R, S, jit, gen.light.param_direction, P, z)
c2b = torch.from_numpy(c2b_h).cuda(); w2b = torch.from_numpy(w2b_h).cuda(); xy = torch.from_numpy(xy_h).cuda()
ro, rd, near, far, ld = ops.gen_rays(c2b, kinv, xy, R, w2b=w2b, light_direction=gen.light.param_direction)
w, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], z=z)
zc, pc = ops.coarse_samples(ro.view(-1, 3), rd.view(-1, 3), near, far, S, jit)
for name, a, b in (("rays_o", pre["rays_o"], ro), ("rays_d", pre["rays_d"], rd), ("near", pre["near"], near), ("far", pre["far"], far),
                   ("light_dir", pre["light_dir"], ld), ("w", pre["w"], w), ("gamma", pre["gamma"], gamma), ("beta", pre["beta"], beta),
                   ("z_coarse", pre["z_coarse"], zc), ("pts_coarse", pre["pts_coarse"], pc)):
    d = (a.reshape(-1) - b.reshape(-1)).abs()
    print(f"{name:12s} equal={torch.equal(a.reshape(-1), b.reshape(-1))} maxdiff={float(d.max()):.3e} n_diff={int((d > 0).sum())}")
