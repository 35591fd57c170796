import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import torch, bench
dev = torch.device("cuda")
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", dev)
disc.eval()
for bsz in (1, 4, 1, 2):
    x = torch.rand(bsz, 3, 64, 64, device=dev)
    with torch.no_grad():
        for rep in range(3):
            ms = bench._ev_time(lambda: disc(x, it=0), 30)
            t = time.perf_counter()
            for _ in range(200): disc(x, it=0)
            torch.cuda.synchronize()
            print(bsz, rep, "ev_time ms", round(ms, 4), "wall us/call", round((time.perf_counter() - t) / 200 * 1e6, 1), "plans", "-")
