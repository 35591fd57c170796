"""Launch list of ONE discriminator training step (real fwd + R1 double-backward + fake fwd + bwd + RMSprop) for
`rocprofv3 --kernel-trace`: the step is bracketed by two oi_lrelu_mask_mul launches on a 1-element tensor... no: by
two torch.cuda.nvtx-free markers = zero-size-safe `torch.full((7,), ...)` fills are ambiguous, so the script prints the
wall time and relies on the kernel trace being dominated by the N identical steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
from oi_amd.optim import FusedAdam, FusedRMSprop
from oi_amd.trainer import Trainer
dev = torch.device("cuda")
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", dev)
mods = {"generator": gen, "discriminator": disc, "mask_discriminator": disc,
        "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
        "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4), "opt_mask_discriminator": None}
tr = Trainer(mods, it=0)
data = {"image": torch.rand(1, 3, 64, 64, device=dev), "mask": torch.rand(1, 1, 64, 64, device=dev)}
with torch.no_grad():
    fake = gen(bs=1, it=0, data={})["box"]
fake_d = {**fake["render_out"], "c2b": fake["prior_info"]["c2b"]}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    tr.train_step_discriminator("discriminator", data, fake_d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    tr.train_step_discriminator("discriminator", data, fake_d)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"D step: host {1e3 * (t1 - t0) / N:.3f} ms, host+drain {1e3 * (t2 - t0) / N:.3f} ms")
