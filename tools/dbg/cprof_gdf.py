"""cProfile of the graphed discriminator forward loop at batch 1 (bench d_images_per_s): where the host time goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/object-intrinsics_amd")
import torch
import bench
from oi_amd.graphed import GraphedDForward
dev = torch.device("cuda", 0)
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", dev)
disc.eval()
x = torch.rand(1, 3, 64, 64, device=dev)
gd = GraphedDForward(disc)
for _ in range(20):
    gd(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    gd(x)
torch.cuda.synchronize()
print("us per call (wall):", (time.perf_counter() - t0) / 500 * 1e6)
t0 = time.perf_counter()
for _ in range(500):
    gd.graph.replay()
torch.cuda.synchronize()
print("us per bare replay:", (time.perf_counter() - t0) / 500 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(500):
    gd(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
