"""ATen ops dispatched from Python during steady-state training iterations, with the innermost oi_amd frame (ops of the
autograd thread are not seen by a dispatch mode)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
exec(open(os.path.join(ROOT, "tools", "dbg", "prof_gstep.py")).read().split("N = 3\n")[0])
N = 3
log = Log()
with log:
    for _ in range(N):
        tr.train_step(data)
torch.cuda.synchronize()
print("ops dispatched from Python per iteration (views excluded):", sum(log.c.values()) / N)
for (name, where), n in log.c.most_common(90):
    print(f"{n / N:6.2f}  {name:34s} {where}")
