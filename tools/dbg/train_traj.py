"""Training trajectories of the f16x3 (default) and native-fp32 operand modes from identical seeds: per-iteration losses."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import numpy as np, torch, bench
from oi_amd.config import build_from_config
from oi_amd.trainer import Trainer
from oi_amd.optim import FusedAdam, FusedRMSprop

def run(prec, n):
    torch.manual_seed(0); np.random.seed(0)
    gen, disc = bench.build_models(32, 16, 16, 1, prec, torch.device("cuda"))
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator", aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1),
                                  aug_p=1, img_size=32, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).cuda()
    mods = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
            "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
            "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4), "opt_mask_discriminator": FusedRMSprop(mdisc.parameters(), lr=1e-4)}
    tr = Trainer(mods)
    g = torch.Generator().manual_seed(1)
    out = []
    for it in range(n):
        data = {"image": torch.rand(1, 3, 32, 32, generator=g).cuda(), "mask": (torch.rand(1, 1, 32, 32, generator=g) > 0.5).float().cuda()}
        torch.manual_seed(100 + it); np.random.seed(100 + it)
        o = tr.train_step(data)
        out.append([float(o[k]) for k in ("generator/loss", "generator/eikonal", "discriminator/loss", "discriminator/reg", "mask_discriminator/loss")])
    return np.array(out)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
a, b = run("f32", n), run("f16x3", n)
print("finite:", np.isfinite(a).all(), np.isfinite(b).all())
for it in (0, 1, 5, 10, 20, n - 1):
    print(it, "f32  ", np.round(a[it], 5), "\n  f16x3", np.round(b[it], 5), " maxdiff", float(np.abs(a[it] - b[it]).max()))
