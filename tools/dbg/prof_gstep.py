"""Which ATen ops (and from where) a generator training step launches: torch profiler with stacks over
Trainer.train_step_generator, aggregated by op and the innermost oi_amd / torch.autograd frame."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
from oi_amd.config import build_from_config
from oi_amd.optim import FusedAdam, FusedRMSprop
from oi_amd.trainer import Trainer
from torch.profiler import profile, ProfilerActivity
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
tr = Trainer(mods, graph_d_steps=True)
data = {"image": torch.rand(1, 3, 64, 64, device=dev), "mask": torch.rand(1, 1, 64, 64, device=dev)}
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode


class Log(TorchDispatchMode):
    """every ATen op dispatched from Python (forward; ops of the autograd thread are not seen) with the innermost oi_amd frame"""

    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(k in name for k in ("view", "detach", "alias", "empty", "expand", "reshape", "select", "slice", "t.default",
                                       "transpose", "unsqueeze", "squeeze", "permute", "as_strided", "_local_scalar", "is_", "size")):
            fr = [f for f in traceback.extract_stack() if "oi_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd / torch)"
            self.c[(name, where)] += 1
        return func(*args, **(kwargs or {}))


N = 3
log = Log()
with log:
    for _ in range(N):
        tr.train_step_generator(1)
torch.cuda.synchronize()
print("ops dispatched from Python per generator step (views excluded):", sum(log.c.values()) / N)
for (name, where), n in log.c.most_common(70):
    print(f"{n / N:6.1f}  {name:34s} {where}")


# the same for the CAPTURED discriminator steps: a fresh trainer, whose first train_step captures both graphs under the log
# (what the log shows is what every replay launches, plus the per-step host-side glue around the replays)
if os.environ.get("OI_PROF_DSTEPS", "1") != "0":
    tr2 = Trainer(mods, graph_d_steps=True)
    log2 = Log()
    with log2:
        tr2.train_step(data)
    torch.cuda.synchronize()
    g = sum(n for (name, where), n in log2.c.items())
    print("\nops dispatched during ONE whole first iteration (captures both discriminator steps; includes its generator step):", g)
    for (name, where), n in log2.c.most_common(80):
        if "graphed" in where or "discriminator" in where or "augment" in where or "losses" in where or "autograd" in where:
            print(f"{n:6d}  {name:34s} {where}")
