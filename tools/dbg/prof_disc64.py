"""Launch list of ADADiscriminatorView forwards at B = 64 (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
from oi_amd.config import build_from_config
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
disc = build_from_config(net("src.models.discriminator.ADADiscriminatorView",
                             aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=64,
                             in_dim=3, last_bias=False, n_feat=512, out_dim=7, out_dim_latent=0, out_dim_position=6)).cuda().eval()
B = int(os.environ.get("OI_DBG_B", 64))
x = torch.rand(B, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(8):
        disc(x, it=0)
torch.cuda.synchronize()
