import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests", ROOT + "/oracle"):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_modules import _ada_disc
import oi_amd.discriminator as DM
D = _ada_disc(3, 7).cuda().eval()
x = torch.rand(1, 3, 64, 64, device="cuda")
for small in (True, False):
    DM.SMALL_PATH = small
    with torch.no_grad():
        for _ in range(20):
            D(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            D(x)
        torch.cuda.synchronize()
        print("small" if small else "general", (time.perf_counter() - t) / 200 * 1e6, "us per forward")
