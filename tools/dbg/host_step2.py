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
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(45)
