"""Host time of one forward step (Generator.forward + D forward) against its GPU time, per precision mode."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import torch, bench
prec = os.environ.get("PREC", "bf16")
gen, disc = bench.build_models(64, 64, 64, 1, prec, torch.device("cuda"))
gen.train(); disc.eval()
def step(i):
    with torch.no_grad():
        out = gen(bs=1, it=i, data={})["box"]["render_out"]
        disc(out["image"].contiguous(), it=i)
for i in range(10): step(i)
torch.cuda.synchronize()
N = 300
t = time.perf_counter()
for i in range(N): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{prec}: host returns after {(t1 - t) / N * 1e6:.0f} us per step, GPU done after {(t2 - t) / N * 1e6:.0f} us per step")
pr = cProfile.Profile(); pr.enable()
for i in range(100): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
