import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/object-intrinsics_amd")
import numpy as np, torch
import bench
from oi_amd.config import build_from_config
from oi_amd.optim import FusedAdam, FusedRMSprop
from oi_amd.trainer import Trainer
dev = torch.device("cuda")
R = 64
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
torch.manual_seed(1234); np.random.seed(1234)
gen, disc = bench.build_models(R, 64, 64, 1, "f16x3", dev)
gen.train(); disc.eval()
with torch.no_grad():
    for i in range(5):
        out = gen(bs=1, it=i, data={})["box"]["render_out"]; disc(out["image"].contiguous(), it=i)
mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator", aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1,
                              img_size=R, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).to(dev)
mods = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
        "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
        "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4), "opt_mask_discriminator": FusedRMSprop(mdisc.parameters(), lr=1e-4)}
tr = Trainer(mods, graph_d_steps=True)
data = {"image": torch.rand(1, 3, R, R, device=dev), "mask": torch.rand(1, 1, R, R, device=dev)}
for step in range(24):
    out = tr.train_step(data)
    bad = [k for k, v in out.items() if not bool(torch.isfinite(torch.as_tensor(v)).all())]
    gd = tr._graphed["discriminator"]
    gmax = max(float(p.grad.abs().max()) for p in disc.parameters())
    wmax = max(float(p.abs().max()) for p in disc.parameters())
    print(step, bad[:2], "reg", float(out["discriminator/reg"]), "gmax", gmax, "wmax", wmax, "th_real", gd.th_real.flatten().tolist(), "xr", float(gd.x_real.abs().max()), float(gd.x_fake.abs().max()))
