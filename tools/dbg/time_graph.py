import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_modules import build_generator
from oi_amd.graphed import GraphedForward
for R, S in ((16, 16), (32, 32), (64, 64)):
    gen = build_generator(R, S, S, 1, "f16x3"); gen.eval()
    b2w = torch.tensor(np.asarray(gen.pose_prior(1), dtype=np.float32)).cuda(); z = torch.randn(1, 64).cuda()
    data = {"b2w": b2w, "z": z, "bg_color": torch.zeros(1, 3).cuda()}
    with torch.no_grad():
        for _ in range(5): gen(bs=1, it=0, data=data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): gen(bs=1, it=0, data=data)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 100
    gf = GraphedForward(gen, bs=1, it=0, return_raw=False).recapture()
    for _ in range(5): gf(b2w, z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): gf(b2w, z)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 100
    print(f"{R}x{R}, {S}+{S} samples: eager {te*1e3:.3f} ms/frame, hipGraph {tg*1e3:.3f} ms/frame  ({te/tg:.2f}x)")
