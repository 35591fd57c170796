"""cProfile of the batch-1 discriminator forward loop of bench.py (host overhead per D image)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", torch.device("cuda"))
disc.eval()
x = torch.rand(1, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(20):
        disc(x, it=0)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        disc(x, it=0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host {1e6 * (t1 - t0) / 200:.1f} us per forward, host+drain {1e6 * (t2 - t0) / 200:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        disc(x, it=0)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
