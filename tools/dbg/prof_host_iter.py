import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_host_vs_gpu.py")).read().split("N = 30")[0])
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.train_step(data)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
