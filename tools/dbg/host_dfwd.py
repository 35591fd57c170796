"""Is the replayed discriminator forward host- or GPU-bound?  wall per call vs host time of its pieces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import numpy as np, torch, bench
from oi_amd.graphed import GraphedDForward
dev = torch.device("cuda")
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", dev)
x = torch.rand(1, 3, 64, 64, device=dev)
gd = GraphedDForward(disc)
for _ in range(5): gd(x)
torch.cuda.synchronize()
def wall(fn, n=2000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("gd(x) wall per call           %.1f us" % wall(lambda: gd(x)))
print("thetas only (host numpy)      %.1f us" % wall(lambda: gd._thetas(tuple(x.shape))))
th = gd._thetas(tuple(x.shape))
if gd._lib is not None:
    print("library graph, fixed theta    %.1f us" % wall(lambda: gd._lib(x, th)))
m = disc.aug.static_margins(64, 64)
def eager():
    th_ = gd._thetas(tuple(x.shape))
    return disc._forward_small(x, f12=disc.aug.Hz_geom, theta_np=th_, margins=m)
with torch.no_grad():
    print("eager small path (4 launches)  %.1f us" % wall(eager))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(300): gd._lib(x, th)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("300 library-graph launches: host returns after %.1f us per call, GPU done after %.1f us per call" % ((t1 - t) / 300 * 1e6, (t2 - t) / 300 * 1e6))
gd2 = GraphedDForward(disc)
for _ in range(3): gd2(x)
def alt():
    gd._lib(x, th); gd2._lib(x, th)
print("two graph objects alternating   %.1f us per call" % (wall(alt, 1000) / 2))
