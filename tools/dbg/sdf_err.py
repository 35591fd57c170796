#!/usr/bin/env python3
"""Error of the sdf-only and the full forward pass (library selected by OI_LIB) against the fp64 oracle on random points and
latents: max / mean |sdf - sdf_oracle|, for comparing kernel variants' accuracy (not only their agreement)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests", ROOT + "/oracle"):
    sys.path.insert(0, p)
import torch
import oi_oracle as O
from conftest import load_golden
from oi_amd import ops
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack

kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
sdf_net = ShapeNetwork(os.path.join(ROOT, "tests", "golden", "weights_sdf.npz"), **kw).cuda()
col = ColorNetwork(**kw); col.load_state_dict(load_golden("weights_color")); col = col.cuda()
sdf_sd = {k: v.detach().cpu() for k, v in sdf_net.state_dict().items()}
g = torch.Generator().manual_seed(11)
B, n = 3, 6000
pts = torch.rand(B * n, 3, generator=g) * 2.2 - 1.1
z = torch.randn(B, 64, generator=g)
w = O.style_mlp(sdf_sd, z)
sd64 = {k: v.double() for k, v in sdf_sd.items()}
sdf_o, feat_o, grad_o = O.sdf_forward(sd64, pts.double(), O.style_mlp(sd64, z.double()), want_grad=True)
for mode in sys.argv[1:] or ["f16x3", "f32"]:
    pack = FieldPack(sdf_net, col, mode)
    with torch.no_grad():
        _, gamma, beta = pack.film(w=w.cuda())
        full = ops.sdf_mlp_fwd(pts.cuda(), pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig, True, True, False, None)
        only = ops.sdf_mlp_fwd(pts.cuda(), pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig)
    for name, s in (("sdf-only", only[0]), ("full", full[0])):
        d = (s.double().cpu() - sdf_o.squeeze(-1).double()).abs()
        print(f"{mode:6s} {name:8s} max {d.max().item():.3e}  mean {d.mean().item():.3e}")
    d = (full[1].double().cpu() - grad_o.double()).abs()
    print(f"{mode:6s} gradient max {d.max().item():.3e}  mean {d.mean().item():.3e}  (max |grad| {grad_o.abs().max().item():.2f})")
