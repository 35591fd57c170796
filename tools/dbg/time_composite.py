import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests", ROOT + "/oracle"):
    sys.path.insert(0, p)
import torch, oi_oracle as O
from oi_amd import ops
N, T, B = 4096, 128, 1
g = torch.Generator().manual_seed(0)
ro = (torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)).cuda().contiguous()
rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.15 * torch.randn(N, 3, generator=g), dim=-1).cuda().contiguous()
near, far = O.near_far_from_sphere(ro.cpu(), rd.cpu())
z = torch.sort(near + (far - near) * torch.rand(N, T, generator=g), -1).values.cuda()
dists, mid_z, _ = ops.midpoints(ro, rd, z, 2.0 / T)
sdf = (1.0 - mid_z) * 0.3
grad = torch.nn.functional.normalize(torch.randn(N, T, 3, generator=g), dim=-1).cuda()
rgb = torch.rand(N, T, 3, generator=g).cuda()
args = (sdf, grad, rgb, dists, mid_z, ro, rd, torch.tensor([[0.2, -0.4, -0.9]]).cuda(), torch.rand(B, 3).cuda(), torch.tensor(0.3).cuda(),
        torch.tensor([-0.7, 0.2, 8.0]).cuda(), 0.5, B)
def t(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
outs_gen = ("weights", "weight_sum", "weight_max", "color_fine", "image_no_bg", "image", "shading", "normal", "mask", "z_map", "specular_map", "diffuse_map", "cdf", "reduce4")
for fused in (True, False):
    ops.FUSED_STATS = fused
    for planar in (False, True):
        for outs in (None, outs_gen, ("image", "mask", "reduce4"), ("image", "mask")):
            print(f"fused={fused} planar={planar} outputs={'all' if outs is None else len(outs)}: {t(lambda: ops.composite_fwd(*args, outs, image_planar=planar)):.2f} us")
