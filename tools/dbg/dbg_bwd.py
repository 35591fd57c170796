import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
import oi_oracle as O
from conftest import load_golden
from test_gpu_backward import _oracle_mlp_grads, NET_KW, SDF_NPZ
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
from oi_amd.autograd import sdf_mlp
sdf_sd, col_sd = load_golden("weights_sdf"), load_golden("weights_color")
n, B = 96, 2
g = torch.Generator().manual_seed(n)
pts = torch.rand(B * n, 3, generator=g) * 2.0 - 1.0
w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
cs, cg, cr = torch.randn(B * n, generator=g), 0.1 * torch.randn(B * n, 3, generator=g), torch.randn(B * n, 3, generator=g)
loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr)
sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda(); col_net = ColorNetwork(**NET_KW); col_net.load_state_dict(col_sd); col_net = col_net.cuda()
pack = FieldPack(sdf_net, col_net, sys.argv[1] if len(sys.argv) > 1 else "f32")
wh = w.cuda().requires_grad_(True)
_, gamma, beta = pack.film(w=wh)
gamma.retain_grad(); beta.retain_grad()
sdf, grad, rgb, _ = sdf_mlp(pack, pts.cuda(), gamma, beta, B, True, True, False)
loss = (sdf * cs.cuda()).sum() + (grad * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
loss.backward()
for name, p in list(sdf_net.named_parameters()) + [("col." + k, v) for k, v in col_net.named_parameters()]:
    if name.startswith("style"): continue
    key = ("sdf." + name) if not name.startswith("col.") else name
    a, b = p.grad.cpu().double().flatten(), g_o[key].flatten()
    print(f"{key:45s} rel {float((a-b).abs().max()/b.abs().max()):9.3e}  hip {a[:3].numpy()}  ref {b[:3].numpy()}")
for key, p in (("sdf.pts_linears.0.bias", sdf_net.pts_linears[0].bias), ("sdf.pts_linears.0.beta.bias", sdf_net.pts_linears[0].beta.bias)):
    a, b = p.grad.cpu().double().flatten(), g_o[key].flatten()
    bad = ((a - b).abs() > 1e-3 * b.abs().max()).nonzero().flatten().tolist()
    print(key, "bad idx", bad)
    print("  hip", a[bad][:8].numpy(), "\n  ref", b[bad][:8].numpy())
a = sdf_net.pts_linears[0].weight.grad.cpu().double(); b = g_o["sdf.pts_linears.0.weight"]
bad = ((a - b).abs() > 1e-3 * b.abs().max()).nonzero().tolist(); print("W0 bad", bad[:40])
gb = beta.grad.cpu(); print("d_beta[:,0] per element sums", gb[:, 0].sum(-1), " ref bias-sum", float(g_o["sdf.pts_linears.0.beta.bias"].sum()) / 0.25)
