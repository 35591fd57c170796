import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", torch.device("cuda"))
disc.eval()
x = torch.rand(1, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(3): disc(x, it=0)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50): disc(x, it=0)
    torch.cuda.synchronize()
    print("disc fwd ms", (time.perf_counter() - t0) / 50 * 1e3)
