"""Is a training iteration host-bound or GPU-bound?  Host enqueue time vs enqueue + drain, per phase."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
from oi_amd.config import build_from_config
from oi_amd.optim import FusedAdam, FusedRMSprop
from oi_amd.trainer import Trainer
dev = torch.device("cuda")
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", dev)
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator",
                              aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1,
                              img_size=64, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).to(dev)
mods = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
        "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
        "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4),
        "opt_mask_discriminator": FusedRMSprop(mdisc.parameters(), lr=1e-4)}
tr = Trainer(mods, graph_d_steps=os.environ.get("OI_GRAPH", "1") == "1")  # what bench.py times
data = {"image": torch.rand(1, 3, 64, 64, device=dev), "mask": torch.rand(1, 1, 64, 64, device=dev)}
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    tr.train_step(data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"iteration: host {1e3 * (t1 - t0) / N:.2f} ms, host+drain {1e3 * (t2 - t0) / N:.2f} ms")
# per phase, each drained (no overlap between phases): upper bounds of the two resources
def phase(fn, n=15):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        fn()
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    return 1e3 * (b - a) / n, 1e3 * (c - a) / n
with torch.no_grad():
    fake = gen(bs=1, it=0, data={})["box"]
fd = {**fake["render_out"], "c2b": fake["prior_info"]["c2b"]}
def nograd_render():
    with torch.no_grad():
        gen(bs=1, it=tr.it, data={})
for name, fn in (("G step", lambda: tr.train_step_generator(1)), ("no-grad render", nograd_render),
                 ("D step", lambda: tr.train_step_discriminator("discriminator", data, fd)),
                 ("mask-D step", lambda: tr.train_step_discriminator("mask_discriminator", data, fd))):
    h, t = phase(fn)
    print(f"{name:16s} host {h:6.2f} ms   host+drain {t:6.2f} ms")
if os.environ.get("OI_PROF_GSTEP") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(10):
        tr.train_step_generator(1)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
