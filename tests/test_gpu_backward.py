"""GPU parity of the backward HIP kernels against autograd through the oracle (CPU) and against the
reference's own gradients (golden fixtures F6, F7): first-order parameter gradients of losses that
contain second-order terms (normals / eikonal in the generator, R1 in the discriminators)."""
import math
import os

import numpy as np
import pytest
import torch

import oi_oracle as O
from conftest import GOLDEN, load_golden, maxdiff, sub_sd, record_margin

pytestmark = pytest.mark.gpu
NET_KW = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
SDF_NPZ = os.path.join(GOLDEN, "weights_sdf.npz")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


# ---------------------------------------------------------------------------------------------
# discriminator
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,H,Cout,stride,pad", [(2, 3, 16, 8, 2, 1), (1, 64, 16, 128, 2, 1), (3, 32, 4, 7, 1, 0),
                                                      (2, 5, 9, 6, 2, 1), (1, 256, 8, 512, 2, 1)])
def test_conv_dgrad_wgrad_vs_torch(B, Cin, H, Cout, stride, pad):
    from oi_amd import ops
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g, dtype=torch.float64) / math.sqrt(Cin * 16)).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    dx = ops.conv4x4_dgrad(gy.float().cuda(), w.detach().float().cuda(), H, H, stride, pad)
    dw = ops.conv4x4_wgrad(gy.float().cuda(), x.detach().float().cuda(), stride, pad)
    assert rel_err(dx, gx) < 2e-5 and rel_err(dw, gw) < 2e-5


F7_TOL = 3e-6   # measured 7.0e-7 worst over 13 tensors (tools/grad_margin.py); round 2 accepted 2e-4


@pytest.mark.parametrize("tag,res,nf,cin,cout", [("r16c3_", 16, 32, 3, 7), ("r64c3_", 64, 64, 3, 7), ("r64c1_", 64, 32, 1, 1)])
def test_discriminator_r1_grads_golden_f7(tag, res, nf, cin, cout):
    """loss = BCE(D(x)[:, :1], 1) + 10 * R1: weight gradients need the conv double-backward."""
    from oi_amd.discriminator import DCDiscriminator
    from oi_amd.losses import GANLoss, compute_grad2
    g = load_golden("f7_discriminator")
    D = DCDiscriminator(in_dim=cin, out_dim=cout, n_feat=nf, img_size=res)
    D.load_state_dict(sub_sd(g, tag + "w."))
    D = D.cuda()
    x = g[tag + "x"].cuda().requires_grad_(True)
    d = D(x)
    assert maxdiff(d.cpu(), g[tag + "d"]) < 2e-5
    d1 = d[:, :1]
    reg = compute_grad2(d1, x)
    assert abs(float(reg.detach()) - float(g[tag + "reg"])) < 1e-4 * max(1.0, float(g[tag + "reg"]))
    loss = GANLoss("bce")(d1, 1) + 10.0 * reg
    (gx,) = torch.autograd.grad(d1.sum(), x, retain_graph=True)
    assert maxdiff(gx.cpu(), g[tag + "gx"]) < 1e-5
    gw = torch.autograd.grad(loss, list(D.parameters()))
    for (k, _), gr in zip(D.named_parameters(), gw):
        ref = g[tag + "g." + k]
        record_margin("f7_discriminator_r1_weight_grads_vs_reference", tag + k, rel_err(gr, ref))
        assert rel_err(gr, ref) < F7_TOL, (k, rel_err(gr, ref))


def test_ada_discriminator_backward_vs_oracle():
    """Full ADA pipeline (pad, upfirdn2d x4, grid sample, convs) input-gradient and R1 vs the oracle."""
    from oi_amd.discriminator import ADADiscriminator
    from oi_amd.losses import compute_grad2
    torch.manual_seed(0)
    D = ADADiscriminator(aug={"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"xint": 1, "scale": 1}},
                         aug_p=1, in_dim=3, out_dim=1, n_feat=32, img_size=32, last_bias=False)
    dsd = {k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items() if "aug." not in k}
    x0 = torch.rand(2, 3, 32, 32)
    pct = 0.7
    p = torch.tensor(pct)
    G = O.ada_G_inv(2, 32, 32, ((p * 2 - 1) * 0.125).expand(2, 2), torch.exp2(torch.erfinv(p * 2 - 1) * 0.2).expand(2))
    xo = x0.clone().requires_grad_(True)
    do = O.dc_discriminator(dsd, O.ada_geometric(xo, G)[0])
    rego = O.r1_penalty(do, xo)
    lo = O.bce_logits_const(do, 1) + 10.0 * rego
    gwo = torch.autograd.grad(lo, list(dsd.values()))

    D = D.cuda()
    orig = D.aug.forward
    D.aug.forward = lambda im: orig(im, debug_percentile=pct)
    x = x0.cuda().requires_grad_(True)
    d = D(x)
    assert maxdiff(d.cpu(), do) < 2e-5
    reg = compute_grad2(d, x)
    assert abs(float(reg) - float(rego)) < 1e-4 * max(1.0, float(rego))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(d, torch.ones_like(d)) + 10.0 * reg
    names = [k for k in dsd]
    gw = torch.autograd.grad(loss, [dict(D.named_parameters())[k] for k in names])
    for k, a, b in zip(names, gw, gwo):
        assert rel_err(a, b) < 3e-4, (k, rel_err(a, b))


# ---- the SHIPPED discriminators (configs/train.yaml:78-102): 128 x 128, 3->32->64->128->256->512->7 and the 1-channel mask
# network, ADA on at a pinned percentile.  Inputs of F14 are chosen with a margin to every LeakyReLU kink (oracle/gen_golden_r5.py:
# a sign change of one pre-activation moves every R1 weight gradient by ~1e-3), so the no-flip accuracy can be demanded.
# Bars = 3x the worst error measured on the MI355X (profiles/r5_gradient_margins.txt).  The reference's own fp32 noise at this
# size is 3.6e-6 (fp64 oracle vs the fixture, tests/test_oracle_golden.py), which is why the bar against the FIXTURE cannot be
# the 3e-6 of the 64 x 64 case; against the fp64 oracle only this implementation's error counts.
# measured: weight gradients 5.4e-6 (vs the fixture) / 4.7e-6 (vs the fp64 oracle) worst over 4 cases x 6 tensors, input gradient
# 1.7e-5 / 1.8e-5 (the CPU oracle in fp32 against the fixture: 4.5e-6 and 1.7e-5)
F14_TOL = {"d": 2e-5, "reg": 1e-4, "gx": 5e-5, "gw_ref": 1.6e-5, "gw_oracle": 1.5e-5, "gx_oracle": 5e-5}


def _f14_net(g, tag, B):
    from conftest import f14_weights, F14_NETS
    from oi_amd.config import build_from_config
    kw = F14_NETS[tag]
    aug = {"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}}
    common = dict(aug=aug, aug_p=1, in_dim=kw["in_dim"], out_dim=kw["out_dim"], n_feat=512, img_size=128, last_bias=False)
    if kw["view"]:
        cfg = {"__target__": "src.models.discriminator.ADADiscriminatorView", "kwargs": dict(out_dim_position=6, out_dim_latent=0, **common)}
    else:
        cfg = {"__target__": "src.models.discriminator.ADADiscriminator", "kwargs": common}
    D = build_from_config(cfg)
    wsd = f14_weights(g, tag)
    D.load_state_dict({**{k: v for k, v in D.state_dict().items() if "aug." in k}, **wsd})
    D = D.cuda()
    pct = float(g[f"{tag}b{B}_pct"])
    orig = D.aug.forward
    D.aug.forward = lambda im, **k: orig(im, debug_percentile=pct, **k)
    return D, wsd, pct


@pytest.mark.parametrize("tag,B", [("v_", 1), ("v_", 2), ("m_", 1), ("m_", 2)])
def test_shipped_discriminators_128_golden_f14(tag, B):
    """ADADiscriminatorView / ADADiscriminator at 128 x 128 against the reference's own outputs AND against the fp64 oracle:
    logits (with and without a gradient recorded), R1 penalty, d sum(D[:, :1]) / dx, weight gradients of BCE + 10 R1."""
    from conftest import f14_grad_errors
    from oi_amd.losses import GANLoss, compute_grad2
    g = load_golden("f14_discriminator_128")
    t = f"{tag}b{B}_"
    D, wsd, pct = _f14_net(g, tag, B)
    x = g[t + "x"].cuda().requires_grad_(True)
    with torch.no_grad():
        d0 = D(x.detach())
    d = D(x)
    assert maxdiff(d0.cpu(), g[t + "d"]) < F14_TOL["d"] and maxdiff(d.cpu(), g[t + "d"]) < F14_TOL["d"]
    d1 = d[:, :1]
    reg = compute_grad2(d1, x)
    assert abs(float(reg.detach()) - float(g[t + "reg"])) < F14_TOL["reg"] * max(1.0, float(g[t + "reg"]))
    loss = GANLoss("bce")(d1, 1) + 10.0 * reg
    assert abs(float(loss.detach()) - float(g[t + "loss"])) < F14_TOL["reg"] * max(1.0, float(g[t + "loss"]))
    (gx,) = torch.autograd.grad(d1.sum(), x, retain_graph=True)
    e_gx = rel_err(gx, g[t + "gx"])
    record_margin("f14_discriminator_128_vs_reference", t + "gx", e_gx)
    names = [k for k, _ in D.named_parameters()]
    gw = torch.autograd.grad(loss, [p_ for _, p_ in D.named_parameters()])
    errs = f14_grad_errors(g, t, zip(names, gw))
    for k, e in errs.items():
        record_margin("f14_discriminator_128_r1_weight_grads_vs_reference", t + k, e)
    # ---- the same quantities from the fp64 oracle (O.ada_geometric + O.dc_discriminator, autograd)
    dsd = {k: v.double().clone().requires_grad_(True) for k, v in wsd.items()}
    xo = g[t + "x"].double().clone().requires_grad_(True)
    p = torch.tensor(pct)
    G = O.ada_G_inv(B, 128, 128, ((p * 2 - 1) * 0.125).expand(B, 2), torch.exp2(torch.erfinv(p * 2 - 1) * 0.2).expand(B)).double()
    do = O.dc_discriminator(dsd, O.ada_geometric(xo, G)[0])
    assert maxdiff(d.cpu(), do) < F14_TOL["d"]
    rego = O.r1_penalty(do[:, :1], xo)
    lo = O.bce_logits_const(do[:, :1], 1) + 10.0 * rego
    gwo = torch.autograd.grad(lo, list(dsd.values()), retain_graph=True)
    (gxo,) = torch.autograd.grad(do[:, :1].sum(), xo)
    e_gxo = rel_err(gx, gxo)
    record_margin("f14_discriminator_128_vs_fp64_oracle", t + "gx", e_gxo)
    errs_o = {k: rel_err(a, b) for k, a, b in zip(names, gw, gwo)}
    for k, e in errs_o.items():
        record_margin("f14_discriminator_128_r1_weight_grads_vs_fp64_oracle", t + k, e)
    assert e_gx < F14_TOL["gx"] and e_gxo < F14_TOL["gx_oracle"], (e_gx, e_gxo)
    assert max(errs.values()) < F14_TOL["gw_ref"], errs
    assert max(errs_o.values()) < F14_TOL["gw_oracle"], errs_o


# ---------------------------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------------------------
COMPOSITE_BWD_TOL = 1.2e-3  # the shininess exponent (d/dn x^n = x^n ln x in fp32): measured 3.8e-4; every other input <= 1e-5


def test_composite_backward_vs_oracle(sdf_sd, col_sd):
    from oi_amd import ops
    from oi_amd.autograd_render import CompositeFunction
    g = torch.Generator().manual_seed(5)
    B, H, W, T = 2, 3, 4, 70
    N = B * H * W
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.2 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    z = torch.sort(near + (far - near) * torch.rand(N, T, generator=g), -1).values
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    S = 35
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), 2.0 / S)], -1)
    mid = z + dists * 0.5
    pts = ro[:, None] + rd[:, None] * mid[..., None]
    with torch.no_grad():
        sdf0, feat0, grad0 = O.sdf_forward(sdf_sd, pts.reshape(-1, 3), w, want_grad=True)
        rgb0 = O.color_head(col_sd, feat0, grad0, w)
    lsd = {"param_direction": torch.tensor([0.3, -0.5, -0.8]), "param_ambient": torch.tensor(-0.4),
           "param_specular": torch.tensor(0.35), "param_shininess": torch.tensor(6.0)}
    w2b = torch.eye(4).repeat(B, 1, 1)
    w2b[:, :3, :3] = torch.linalg.qr(torch.randn(B, 3, 3, generator=g)).Q
    bg = torch.rand(B, 3, generator=g)
    car = 0.37
    cot = {k: torch.randn(s, generator=g) for k, s in
           (("image", (B, 3, H, W)), ("mask", (B, 1, H, W)), ("shading_map", (B, 3, H, W)), ("normal_map", (B, 3, H, W)),
            ("z_map", (B, 1, H, W)), ("color_map", (B, 3, H, W)), ("weights", (N, T)), ("specular_map", (B, 3, H, W)),
            ("diff_shading_map", (B, 3, H, W)))}

    # ---- oracle (fp64 autograd on CPU)
    dd = lambda t: t.double().clone().requires_grad_(True)
    sdf_o, grad_o, rgb_o, var_o = dd(sdf0.reshape(N, T)), dd(grad0.reshape(N, T, 3)), dd(rgb0.reshape(N, T, 3)), dd(torch.tensor(0.3))
    lsd_o = {k: dd(v) for k, v in lsd.items()}
    inv_s = O.inv_s_from_variance(var_o)
    true_cos = (rd.double()[:, None, :] * grad_o).sum(-1)
    ic = -(torch.relu(-true_cos * 0.5 + 0.5) * (1 - car) + torch.relu(-true_cos) * car)
    dO = dists.double()
    pc = torch.sigmoid((sdf_o - ic * dO * 0.5) * inv_s)
    nc = torch.sigmoid((sdf_o + ic * dO * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clamp(0, 1)
    wts = O.transmittance_weights(alpha)
    pn = torch.linalg.norm(pts.double(), dim=-1)
    relax = (pn < 1.2).double()
    eik = (relax * (torch.linalg.norm(grad_o, dim=-1) - 1) ** 2).sum() / (relax.sum() + 1e-5)
    ro_dict = {"pts": pts.double(), "weights": wts, "weight_sum": wts.sum(-1, keepdim=True), "gradients": grad_o,
               "raw_color": rgb_o, "color_fine": (rgb_o * wts[..., None]).sum(1), "mid_z_vals": mid.double()}
    maps = O.render_maps(ro_dict, ro.double(), lsd_o, w2b.double(), bg.double(), B, H, W, return_raw=True)
    loss_o = sum((maps[k] * cot[k].double()).sum() for k in cot if k != "weights") + (wts * cot["weights"].double()).sum() + 7.0 * eik
    leaves = [sdf_o, grad_o, rgb_o, var_o, lsd_o["param_ambient"], lsd_o["param_specular"], lsd_o["param_shininess"],
              lsd_o["param_direction"]]
    g_o = torch.autograd.grad(loss_o, leaves)

    # ---- HIP
    c = lambda t: t.float().cuda().requires_grad_(True)
    sdf_h, grad_h, rgb_h, var_h = c(sdf0.reshape(N, T)), c(grad0.reshape(N, T, 3)), c(rgb0.reshape(N, T, 3)), c(torch.tensor(0.3))
    lp = {k: c(v) for k, v in lsd.items()}
    light = torch.stack([lp["param_ambient"], lp["param_specular"], lp["param_shininess"]])
    dirn = lp["param_direction"] / torch.linalg.norm(lp["param_direction"])
    ldir = torch.einsum("bij,j->bi", w2b.cuda()[:, :3, :3], dirn)
    out = CompositeFunction.run(sdf_h, grad_h, rgb_h, dists.cuda(), mid.cuda(), ro.cuda(), rd.cuda(), ldir, bg.cuda(),
                                var_h, light, car, B)
    to_map = lambda x: x.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    m = {"image": to_map(out["image"]), "mask": to_map(out["mask"]), "shading_map": to_map(out["shading"]).expand(B, 3, H, W),
         "normal_map": to_map(out["normal"]), "z_map": to_map(out["z_map"]), "color_map": to_map(out["color_fine"]),
         "specular_map": to_map(out["specular_map"]).expand(B, 3, H, W),
         "diff_shading_map": to_map(out["diffuse_map"]).expand(B, 3, H, W)}
    eik_h = out["reduce4"][0] / (out["reduce4"][1] + 1e-5)
    loss_h = sum((m[k] * cot[k].cuda()).sum() for k in m) + (out["weights"] * cot["weights"].cuda()).sum() + 7.0 * eik_h
    assert abs(float(loss_h) - float(loss_o)) < 1e-3 * max(1.0, abs(float(loss_o)))
    g_h = torch.autograd.grad(loss_h, [sdf_h, grad_h, rgb_h, var_h, lp["param_ambient"], lp["param_specular"],
                                       lp["param_shininess"], lp["param_direction"]])
    for name, a, b in zip(("sdf", "grad", "rgb", "variance", "ambient", "specular", "shininess", "direction"), g_h, g_o):
        record_margin("composite_backward_vs_fp64_oracle", name, rel_err(a, b))
        assert rel_err(a, b) < COMPOSITE_BWD_TOL, (name, rel_err(a, b), a.flatten()[:4], b.flatten()[:4])
    # the map layout (image written / its gradient read as (B, 3, H W): what Generator.forward uses) is the same arithmetic:
    # bit-identical image and gradients; with a gradient recorded the launch's own logging reductions come along
    out_p = CompositeFunction.run(sdf_h, grad_h, rgb_h, dists.cuda(), mid.cuda(), ro.cuda(), rd.cuda(), ldir, bg.cuda(),
                                  var_h, light, car, B, image_planar=True)
    assert tuple(out_p["image"].shape) == (B, 3, H * W) and torch.equal(out_p["image"].view(B, 3, H, W), m["image"])
    assert not out_p["finals"].requires_grad and torch.equal(out_p["finals"], out["finals"])
    assert abs(float(out_p["finals"][0]) - float(eik_h)) <= 1e-6 * abs(float(eik_h))
    out_q = CompositeFunction.run(sdf_h, grad_h, rgb_h, dists.cuda(), mid.cuda(), ro.cuda(), rd.cuda(), ldir, bg.cuda(),
                                  var_h, light, car, B)   # (a fresh graph: the first one was consumed above)
    loss_p = (out_p["image"].view(B, 3, H, W) * cot["image"].cuda()).sum()
    loss_q = (to_map(out_q["image"]) * cot["image"].cuda()).sum()
    leaves_h = [sdf_h, grad_h, rgb_h, var_h, lp["param_ambient"], lp["param_specular"], lp["param_shininess"]]
    for a, b in zip(torch.autograd.grad(loss_p, leaves_h), torch.autograd.grad(loss_q, leaves_h)):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------------------------
def _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr):
    sd = {k: v.double().clone().requires_grad_(True) for k, v in sdf_sd.items()}
    csd = {k: v.double().clone().requires_grad_(True) for k, v in col_sd.items()}
    wd = w.double().clone().requires_grad_(True)
    sdf, feat, grad = O.sdf_forward(sd, pts.double(), wd, want_grad=True)
    rgb = O.color_head(csd, feat, grad, wd)
    loss = (sdf.squeeze(-1) * cs.double()).sum() + (grad * cg.double()).sum() + (rgb * cr.double()).sum()
    names = [("sdf." + k, v) for k, v in sd.items() if not k.startswith("style.")] + [("col." + k, v) for k, v in csd.items()] + [("w", wd)]
    gr = torch.autograd.grad(loss, [v for _, v in names])
    return float(loss), {n: g_ for (n, _), g_ in zip(names, gr)}


# tolerances: 3x the error measured in the native-fp32 mode (6.7e-6 worst over 59 tensors, tools/grad_margin.py; DESIGN.md
# section 5); bf16x3 (bf16x3 forward, f16x3 backward) 3x its own 1.2e-5
# bf16 (BASELINE configs[1]: one bf16 MFMA per product forward AND backward, unreduced v_sin / v_cos): the bar is 3x the worst
# error measured on the MI355X AT THE FINAL KERNELS (profiles/r6_gradient_margins.txt, row mlp_backward_vs_fp64_oracle[bf16]),
# relative to the largest entry of each tensor
# measured: worst 4.11e-2 (layer-0 FiLM rows), median 1.82e-2 over 60 tensors, loss 4.4e-4 -- bf16 operands (2^-8) under FiLM
# scales of ~30 through eight layers; the fp32-class modes sit at 5e-6
BF16_MLP_BWD_TOL = 0.12
BF16_MLP_BWD_MEDIAN_TOL = 0.05
BF16_LOSS_TOL = 1.3e-3


@pytest.mark.parametrize("n,B,precision,tol", [(96, 2, "f32", 2e-5), (300, 1, "f32", 2e-5), (64, 2, "bf16x3", 4e-5),
                                               (160, 1, "bf16x6", 2e-5), (160, 1, "f16x3", 2e-5),
                                               (160, 1, "bf16", BF16_MLP_BWD_TOL), (96, 2, "bf16", BF16_MLP_BWD_TOL)])
def test_mlp_backward_vs_oracle(sdf_sd, col_sd, n, B, precision, tol):
    """dL/d(every parameter, w) for L = <cs, sdf> + <cg, d sdf/dx> + <cr, rgb> (random cotangents)."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    from oi_amd.autograd import sdf_mlp
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(B * n, 3, generator=g) * 2.0 - 1.0
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    cs, cg, cr = torch.randn(B * n, generator=g), 0.1 * torch.randn(B * n, 3, generator=g), torch.randn(B * n, 3, generator=g)
    loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr)

    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, precision)
    wh = w.cuda().requires_grad_(True)
    _, gamma, beta = pack.film(w=wh)
    sdf, grad, rgb, _ = sdf_mlp(pack, pts.cuda(), gamma, beta, B, True, True, False)
    loss = (sdf * cs.cuda()).sum() + (grad * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
    if precision == "bf16":
        record_margin("mlp_backward_vs_fp64_oracle[bf16]", "(loss)", abs(float(loss) - loss_o) / max(1.0, abs(loss_o)))
    assert abs(float(loss) - loss_o) < (BF16_LOSS_TOL if precision == "bf16" else 1e-3) * max(1.0, abs(loss_o))
    named = [("sdf." + k, v) for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + \
            [("col." + k, v) for k, v in col_net.named_parameters()] + [("w", wh)]
    gr = torch.autograd.grad(loss, [v for _, v in named])
    worst = {}
    for (name, _), a in zip(named, gr):
        worst[name] = rel_err(a, g_o[name])
        record_margin(f"mlp_backward_vs_fp64_oracle[{precision}]", name, worst[name])
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, bad
    if precision == "bf16":
        import statistics
        assert statistics.median(worst.values()) < BF16_MLP_BWD_MEDIAN_TOL, statistics.median(worst.values())


# measured 1.5e-6 worst (d normals; every other tensor <= 6e-7): exact-fp32 MFMA products, fixed summation order
COLOR_HEAD_BWD_TOL = 5e-6


@pytest.mark.parametrize("npe,B", [(200, 2), (1, 1), (4100, 1)])
def test_color_network_standalone_backward_vs_oracle(col_sd, npe, B):
    """d L / d (feature_vectors, normals, w, every parameter) of ColorNetwork.forward (oi_color_head_bwd) for L = <c, rgb>
    against fp64 autograd through the oracle (fields.py:89-101); bit-reproducible (no atomics)."""
    from oi_amd.fields import ColorNetwork
    gen = torch.Generator().manual_seed(npe + B)
    n = B * npe
    feat = torch.rand(n, 128, generator=gen) * 2 - 1
    nrm = torch.randn(n, 3, generator=gen) * 3
    w = torch.randn(B, 64, generator=gen)
    c = torch.randn(n, 3, generator=gen)
    csd = {k: v.double().clone().requires_grad_(True) for k, v in col_sd.items()}
    fo, no, wo = (t.double().clone().requires_grad_(True) for t in (feat, nrm, w))
    lo = (O.color_head(csd, fo, no, wo) * c.double()).sum()
    g_o = dict(zip(["feat", "normals", "w"] + list(csd), torch.autograd.grad(lo, [fo, no, wo] + list(csd.values()))))
    col = ColorNetwork(**NET_KW)
    col.load_state_dict(col_sd)
    col = col.cuda()
    fh, nh, wh = (t.cuda().requires_grad_(True) for t in (feat, nrm, w))
    params = dict(col.named_parameters())

    def run():
        rgb = col(torch.zeros(n, 3, device="cuda"), nh, None, fh, None, wh)
        return torch.autograd.grad((rgb * c.cuda()).sum(), [fh, nh, wh] + list(params.values()))

    g_h = dict(zip(["feat", "normals", "w"] + list(params), run()))
    bad = {}
    for k, a in g_h.items():
        e = rel_err(a, g_o[k])
        record_margin("color_head_standalone_backward_vs_fp64_oracle", k, e)
        if e > COLOR_HEAD_BWD_TOL:
            bad[k] = e
    assert set(g_h) == set(g_o) and not bad, bad
    for a, b in zip(g_h.values(), run()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,B", [(150, 2), (700, 1)])
def test_reference_style_field_chain_backward_vs_oracle(sdf_sd, col_sd, n, B):
    """The call pattern of the reference's renderer.py:241-261 on the drop-in modules, un-fused:
        out = sdf_network(pts, z, w); gradients = sdf_network.gradient(pts, z, w); rgb = color_network(pts, gradients, dirs, out[:, 1:], z, w)
    The albedo head's gradient with respect to the FEATURES re-enters the SDF network's backward (oi_sdf_mlp_bwd_feat), its
    gradient with respect to the normals the double backward of `gradient`.  Every parameter and w against fp64 autograd of
    the same loss through the oracle (the loss of test_mlp_backward_vs_oracle: the fused path must agree with this one)."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(B * n, 3, generator=g) * 2.0 - 1.0
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    cs, cg, cr = torch.randn(B * n, generator=g), 0.1 * torch.randn(B * n, 3, generator=g), torch.randn(B * n, 3, generator=g)
    loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr)
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    wh = w.cuda().requires_grad_(True)
    x = pts.cuda()
    out = sdf_net(x, None, wh)
    assert out.shape == (B * n, 129)
    grads = sdf_net.gradient(x, None, wh)
    rgb = col_net(x, grads, None, out[:, 1:], None, wh)
    loss = (out[:, 0] * cs.cuda()).sum() + (grads * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
    assert abs(float(loss) - loss_o) < 1e-3 * max(1.0, abs(loss_o))
    named = [("sdf." + k, v) for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + \
            [("col." + k, v) for k, v in col_net.named_parameters()] + [("w", wh)]
    gr = torch.autograd.grad(loss, [v for _, v in named])
    bad = {}
    for (name, _), a in zip(named, gr):
        e = rel_err(a, g_o[name])
        record_margin("reference_style_field_chain_backward_vs_fp64_oracle", name, e)
        if e > 2e-5:
            bad[name] = e
    assert not bad, bad


# 3x the error measured in the native-fp32 mode against the reference's own (fp32) gradients: 2.3e-5 (F6), 3.5e-5 (F9)
# worst over 69 tensors (tools/grad_margin.py); round 2 accepted 3e-3
F9_D_TOL = 3e-6  # discriminator weight gradients of the D / mask-D steps: measured 9.2e-7; round 2 accepted 2e-3
F6_TOL = 7e-5
F9_TOL = 1e-4
# bf16 operand mode (BASELINE configs[1]) end to end: forward maps AND every gradient of the training losses against the
# reference's own fp32 values.  Bars = ~3x the worst error measured on the MI355X (record_margin ->
# profiles/r6_gradient_margins.txt, re-measured at the final kernels); gradient errors are relative to the largest entry of the
# tensor (floor as in the fp32 rows).
# measured: F6 image 1.0e-3 / mask 1.53e-3 / eikonal 1.34e-4 / loss (a sum over 64 rays x 7 channels) 1.2e-2, gradients 3.41e-2
# worst, 7.6e-3 median; F9 G-step losses 8.8e-4, gradients 4.8e-2 worst; D-step losses 7.7e-5, weight gradients 3.95e-4 (their
# only bf16 input is the fake image)
BF16_F6 = {"image": 3e-3, "mask": 4.6e-3, "eikonal": 4e-4, "loss": 3.6e-2, "grad": 0.10}
BF16_F9 = {"g_loss": 2.7e-3, "g_grad": 0.13, "d_loss": 2e-4, "d_grad": 1.2e-3}


def _f6_render(precision="f16x3"):
    """The F6 forward (reference weights, rays, latent, background, jitter draw) through the HIP path:
    returns the golden dict, image / shading / mask maps, the eikonal term and the named generator parameters."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, SingleVarianceNetwork
    from oi_amd.renderer import NeuSRenderer
    from oi_amd.lighting import DirectionalLightWithSpecularFixInit
    g = load_golden("f6_grads")
    p = sub_sd(g, "p.")
    sdf = ShapeNetwork(None, **NET_KW)
    sdf.load_state_dict(sub_sd(p, "sdf_network."))
    col = ColorNetwork(**NET_KW)
    col.load_state_dict(sub_sd(p, "color_network."))
    dev = SingleVarianceNetwork(0.3)
    dev.load_state_dict(sub_sd(p, "deviation_network."))
    light = DirectionalLightWithSpecularFixInit(direction=[0, 0, -1.0])
    light.load_state_dict(sub_sd(p, "light."))
    sdf, col, dev, light = sdf.cuda(), col.cuda(), dev.cuda(), light.cuda()
    r = NeuSRenderer(None, sdf, dev, col, n_samples=8, n_importance=8, n_outside=0, up_sample_steps=1, perturb=1,
                     precision=precision)
    ro, rd = g["rays_o"].cuda(), g["rays_d"].cuda()
    near, far = O.near_far_from_sphere(g["rays_o"], g["rays_d"])
    real_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: g["jitter"].cuda()   # the reference's jitter draw (renderer.py:372)
        w = sdf.style(g["z"].cuda())
        w2b = O.invert_rot_t(g["b2w"]).cuda()
        s, c = r.render_full(ro, rd, near.cuda(), far.cuda(), perturb_overwrite=1, cos_anneal_ratio=float(g["cos_anneal_ratio"]),
                             z=g["z"].cuda(), w=w, light=light.packed(), light_dir=light.batch_direction(w2b), bg=g["bg"].cuda())
    finally:
        torch.rand = real_rand
    to_map = lambda x: x.reshape(1, 8, 8, -1).permute(0, 3, 1, 2)
    named = [("sdf_network." + k, v) for k, v in sdf.named_parameters()] + [("color_network." + k, v) for k, v in col.named_parameters()] + \
            [("deviation_network." + k, v) for k, v in dev.named_parameters()] + [("light." + k, v) for k, v in light.named_parameters()]
    return g, to_map(c["image"]), to_map(c["shading"]).expand(1, 3, 8, 8), to_map(c["mask"]), c["reduce4"][0] / (c["reduce4"][1] + 1e-5), named


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16"])
def test_scripted_train_step_golden_f9(precision):
    """F9: one training iteration assembled from the reference's own pieces (gan_pose_trainer.py:103-200 call pattern,
    configs/train.yaml loss weights): G-step loss + generator gradients through both discriminators, D / mask-D step
    losses (real, fake, R1, auxiliary pose regression) + their weight gradients."""
    from oi_amd.config import build_from_config
    from oi_amd.losses import GANLoss, PositionLoss, compute_grad2, linear_increase
    g6, image, _, mask, eik, named = _f6_render(precision)
    g = load_golden("f9_train_step")
    it = int(g["it"])
    aug = {"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}}
    D = build_from_config({"__target__": "src.models.discriminator.ADADiscriminatorView", "kwargs": dict(
        out_dim_position=6, out_dim_latent=0, aug=aug, aug_p=0.0, in_dim=3, out_dim=7, n_feat=32, img_size=8, last_bias=False)})
    M = build_from_config({"__target__": "src.models.discriminator.ADADiscriminator", "kwargs": dict(
        aug=aug, aug_p=0.0, in_dim=1, out_dim=1, n_feat=32, img_size=8, last_bias=False)})
    D.load_state_dict(sub_sd(g, "d_w."))
    M.load_state_dict(sub_sd(g, "m_w."))
    D, M = D.cuda(), M.cuda()
    gan, pos = GANLoss("bce"), PositionLoss("mse")
    # ---- G step
    ld = gan(D(image, it=it)[:, :1], 1)
    lm = gan(M(mask, it=it), 1)
    lg = ld * 1.0 + lm * 0.1 + 10.0 * eik
    bf = precision == "bf16"
    for name, a, b in (("disc", ld, g["g_loss_disc"]), ("mask", lm, g["g_loss_mask"]), ("total", lg, g["g_loss"])):
        if bf:
            record_margin("f9_g_step_losses_vs_reference[bf16]", name, abs(float(a) - float(b)) / max(1.0, abs(float(b))))
        assert abs(float(a) - float(b)) < (BF16_F9["g_loss"] if bf else 1e-4) * max(1.0, abs(float(b))), (name, float(a), float(b))
    grads = torch.autograd.grad(lg, [v for _, v in named], allow_unused=True, retain_graph=True)
    checked, bad = 0, {}
    for (name, _), gr in zip(named, grads):
        key = "gg." + name
        if key not in g:
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        err = maxdiff(gr.cpu(), g[key]) / max(1e-3, float(g[key].abs().max()))
        record_margin(f"f9_g_step_grads_vs_reference[{precision}]", name, err)
        if err > (BF16_F9["g_grad"] if bf else F9_TOL):
            bad[name] = err
        checked += 1
    assert checked > 60 and not bad, bad
    # ---- D step and mask-D step
    for tag, net, xr, xf, aux in (("d", D, g["in_x_real"], image, True), ("m", M, g["in_m_real"], mask, False)):
        xr = xr.cuda().clone().requires_grad_()
        d_real = net(xr, it=it)[:, :1]
        l_real = gan(d_real, 1)
        l_reg = compute_grad2(d_real, xr)
        xf = xf.detach().clone().requires_grad_()
        d_fake = net(xf, it=it)
        l_aux = torch.zeros((), device="cuda")
        if aux:
            d_fake, d_aux = torch.split(d_fake, (1, 6), dim=1)
            l_aux = pos(d_aux, g["c2b"].cuda()[..., :2, :3].flatten(-2, -1))
        l_fake = gan(d_fake, 0)
        loss = l_real + l_fake + l_reg * 10.0 + l_aux * linear_increase(1000, 1)(it)
        for nm, a in (("real", l_real), ("fake", l_fake), ("reg", l_reg), ("aux", l_aux), ("loss", loss)):
            b = float(g[f"{tag}_{nm}"])
            if bf:   # (the fake batch is the bf16-mode render: its error is the image's)
                record_margin("f9_d_step_losses_vs_reference[bf16]", f"{tag}.{nm}", abs(float(a) - b) / max(1.0, abs(b)))
            assert abs(float(a) - b) < (BF16_F9["d_loss"] if bf else 2e-4) * max(1.0, abs(b)), (tag, nm, float(a), b)
        gw = torch.autograd.grad(loss, list(net.parameters()))
        for (k, _), gr in zip(net.named_parameters(), gw):
            ref = g[f"{tag}_g." + k]
            record_margin(f"f9_d_step_weight_grads_vs_reference[{precision}]", f"{tag}.{k}",
                          maxdiff(gr.cpu(), ref) / max(1e-3, float(ref.abs().max())))
            assert maxdiff(gr.cpu(), ref) < (BF16_F9["d_grad"] if bf else F9_D_TOL) * max(1e-3, float(ref.abs().max())), (tag, k, maxdiff(gr.cpu(), ref), float(ref.abs().max()))


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16"])
def test_generator_grads_golden_f6(precision):
    """The reference's own parameter gradients for loss = sum(image) + 10*eikonal + sum(shading) + 0.5*sum(mask)
    (training mode: jitter on, cos_anneal 0.4) -- exercises MLP double-backward + compositing backward + light."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, SingleVarianceNetwork
    from oi_amd.renderer import NeuSRenderer
    from oi_amd.lighting import DirectionalLightWithSpecularFixInit
    g = load_golden("f6_grads")
    p = sub_sd(g, "p.")
    sdf = ShapeNetwork(None, **NET_KW)
    sdf.load_state_dict(sub_sd(p, "sdf_network."))
    col = ColorNetwork(**NET_KW)
    col.load_state_dict(sub_sd(p, "color_network."))
    dev = SingleVarianceNetwork(0.3)
    dev.load_state_dict(sub_sd(p, "deviation_network."))
    light = DirectionalLightWithSpecularFixInit(direction=[0, 0, -1.0])
    light.load_state_dict(sub_sd(p, "light."))
    sdf, col, dev, light = sdf.cuda(), col.cuda(), dev.cuda(), light.cuda()
    r = NeuSRenderer(None, sdf, dev, col, n_samples=8, n_importance=8, n_outside=0, up_sample_steps=1, perturb=1,
                     precision=precision)
    ro, rd = g["rays_o"].cuda(), g["rays_d"].cuda()
    near, far = O.near_far_from_sphere(g["rays_o"], g["rays_d"])
    # inject the reference's jitter draw (first RNG call of render, renderer.py:372)
    import oi_amd.renderer as RR
    real_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: g["jitter"].cuda()
        w = sdf.style(g["z"].cuda())
        w2b = O.invert_rot_t(g["b2w"]).cuda()
        s, c = r.render_full(ro, rd, near.cuda(), far.cuda(), perturb_overwrite=1, cos_anneal_ratio=float(g["cos_anneal_ratio"]),
                             z=g["z"].cuda(), w=w, light=light.packed(), light_dir=light.batch_direction(w2b), bg=g["bg"].cuda())
    finally:
        torch.rand = real_rand
    to_map = lambda x: x.reshape(1, 8, 8, -1).permute(0, 3, 1, 2)
    image, shading, mask = to_map(c["image"]), to_map(c["shading"]).expand(1, 3, 8, 8), to_map(c["mask"])
    eik = c["reduce4"][0] / (c["reduce4"][1] + 1e-5)
    bf = precision == "bf16"
    loss = image.sum() + 10.0 * eik + shading.sum() + 0.5 * mask.sum()
    fwd = {"image": maxdiff(image.cpu(), g["image"]), "mask": maxdiff(mask.cpu(), g["mask"]),
           "eikonal": abs(float(eik) - float(g["eikonal"])), "loss": abs(float(loss) - float(g["loss"]))}
    for k, e in fwd.items():
        if bf:
            record_margin("f6_forward_bf16_mode_vs_reference", k, e)
        assert e < (BF16_F6[k] if bf else {"loss": 1e-3}.get(k, 1e-4)), (k, e)
    named = [("sdf_network." + k, v) for k, v in sdf.named_parameters()] + [("color_network." + k, v) for k, v in col.named_parameters()] + \
            [("deviation_network." + k, v) for k, v in dev.named_parameters()] + [("light." + k, v) for k, v in light.named_parameters()]
    grads = torch.autograd.grad(loss, [v for _, v in named], allow_unused=True)
    checked, bad = 0, {}
    for (name, _), gr in zip(named, grads):
        key = "g." + name
        if key not in g:
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        ref = g[key]
        err = maxdiff(gr.cpu(), ref) / max(1.0, float(ref.abs().max()))
        record_margin(f"f6_generator_grads_vs_reference[{precision}]", name, err)
        if err > (BF16_F6["grad"] if bf else F6_TOL):
            bad[name] = err
        checked += 1
    assert checked > 60 and not bad, bad


def test_full_training_iteration_runs_and_learns():
    """Three optimiser steps through G / D / mask-D (reference call pattern); all losses finite, every
    trainable parameter receives a gradient in its own phase, parameters move."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_modules import build_generator
    from oi_amd.config import build_from_config
    from oi_amd.trainer import Trainer
    torch.manual_seed(0)
    np.random.seed(0)
    R = 16
    gen = build_generator(R, 8, 8, 1)
    mk = lambda cin, cout, cls, extra: build_from_config({"__target__": "src.models.discriminator." + cls, "kwargs": dict(
        aug={"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}}, aug_p=1,
        img_size=R, in_dim=cin, last_bias=False, n_feat=64, out_dim=cout, **extra)}).cuda()
    disc = mk(3, 7, "ADADiscriminatorView", dict(out_dim_latent=0, out_dim_position=6))
    mdisc = mk(1, 1, "ADADiscriminator", {})
    modules = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
               "opt_generator": torch.optim.Adam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
               "opt_discriminator": torch.optim.RMSprop(disc.parameters(), lr=1e-4),
               "opt_mask_discriminator": torch.optim.RMSprop(mdisc.parameters(), lr=1e-4)}
    tr = Trainer(modules)
    before = {k: torch.cat([p.detach().reshape(-1).clone() for p in modules[k].parameters()]) for k in ("generator", "discriminator", "mask_discriminator")}
    data = {"image": torch.rand(2, 3, R, R, device="cuda"), "mask": torch.rand(2, 1, R, R, device="cuda")}
    for _ in range(2):
        out = tr.train_step(data)
    for k, v in out.items():
        assert math.isfinite(float(v)), k
    for k in before:
        after = torch.cat([p.detach().reshape(-1) for p in modules[k].parameters()])
        assert torch.isfinite(after).all()
        assert float((after - before[k]).abs().max()) > 0, k


@pytest.mark.parametrize("from_z", [True, False])
def test_film_params_backward_vs_oracle(sdf_sd, col_sd, from_z):
    """a1/a2 backward kernels (style MLP + 2 x 9 FiLM heads) vs fp64 autograd through the oracle's restatement."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    g = torch.Generator().manual_seed(11)
    B = 3
    src = torch.randn(B, 64, generator=g)
    cg, cb, cw = torch.randn(B, 9, 128, generator=g), torch.randn(B, 9, 128, generator=g), torch.randn(B, 64, generator=g)
    # oracle, fp64
    sd = {k: v.double().requires_grad_() for k, v in sdf_sd.items()}
    csd = {k: v.double().requires_grad_() for k, v in col_sd.items()}
    xs = src.double().requires_grad_()
    wo = O.style_mlp(sd, xs) if from_z else xs
    gs, bs = [], []
    for l in range(8):
        ga, be = O.film_params(sd, f"pts_linears.{l}.", wo)
        gs.append(ga); bs.append(be)
    ga, be = O.film_params(csd, "views_linears.", wo)
    gs.append(ga); bs.append(be)
    loss_o = (torch.stack(gs, 1) * cg.double()).sum() + (torch.stack(bs, 1) * cb.double()).sum() + (wo * cw.double()).sum()
    loss_o.backward()
    # HIP
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW); col_net.load_state_dict(col_sd); col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, "f16x3")
    xh = src.cuda().requires_grad_()
    w, gamma, beta = pack.film(z=xh) if from_z else pack.film(w=xh)
    loss = (gamma * cg.cuda()).sum() + (beta * cb.cuda()).sum() + (w * cw.cuda()).sum()
    loss.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))

    def chk(a, b, name):
        b = b.float()
        assert maxdiff(a.cpu(), b) < 2e-5 * max(1.0, float(b.abs().max())), (name, maxdiff(a.cpu(), b), float(b.abs().max()))

    chk(xh.grad, xs.grad, "input")
    for name, p in sdf_net.named_parameters():
        if ".gamma." in name or ".beta." in name or (from_z and name.startswith("style.")):
            chk(p.grad, sd[name].grad, name)
        elif name.startswith("style."):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
    for name, p in col_net.named_parameters():
        if ".gamma." in name or ".beta." in name:
            chk(p.grad, csd[name].grad, "col." + name)


def test_style_mlp_under_autograd_runs_on_hip_kernels(sdf_sd):
    """`ShapeNetwork.style(z)` with gradients (the reference's own call site, generator.py:237) goes through
    oi_film_params / oi_film_params_bwd(NL = 0), not through ATen matmuls: values and gradients vs fp64 autograd through
    the oracle, and no `mm` / `addmm` / `leaky_relu` kernel in the autograd graph."""
    from oi_amd.fields import ShapeNetwork
    g = torch.Generator().manual_seed(13)
    z, cw = torch.randn(4, 64, generator=g), torch.randn(4, 64, generator=g)
    sd = {k: v.double().requires_grad_() for k, v in sdf_sd.items()}
    zs = z.double().requires_grad_()
    (O.style_mlp(sd, zs) * cw.double()).sum().backward()
    net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    zh = z.cuda().requires_grad_()
    w = net.style(zh)
    assert type(w.grad_fn).__name__ == "StyleFunctionBackward", type(w.grad_fn).__name__
    (w * cw.cuda()).sum().backward()
    assert maxdiff(w.detach().cpu(), O.style_mlp(sdf_sd, z)) < 1e-5
    assert maxdiff(zh.grad.cpu(), zs.grad.float()) < 2e-5 * max(1.0, float(zs.grad.abs().max()))
    for name, p in net.named_parameters():
        if name.startswith("style."):
            ref = sd[name].grad.float()
            assert maxdiff(p.grad.cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max())), name


TRAJ_KEYS = ("generator/loss", "generator/eikonal", "discriminator/loss", "discriminator/reg", "mask_discriminator/loss")


def _trajectory(prec, iters=6, R=16):
    """`iters` full training iterations (G / D / mask-D steps, fused optimisers) at R x R, 8 + 8 samples, from fixed seeds with the
    generator in operand mode `prec`: the logged losses, one row per iteration (TRAJ_KEYS)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_modules import build_generator
    from oi_amd.config import build_from_config
    from oi_amd.optim import FusedAdam, FusedRMSprop
    from oi_amd.trainer import Trainer
    torch.manual_seed(0)
    np.random.seed(0)
    gen = build_generator(R, 8, 8, 1, prec)
    mk = lambda cin, cout, cls, extra: build_from_config({"__target__": "src.models.discriminator." + cls, "kwargs": dict(
        aug={"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}}, aug_p=1,
        img_size=R, in_dim=cin, last_bias=False, n_feat=64, out_dim=cout, **extra)}).cuda()
    disc = mk(3, 7, "ADADiscriminatorView", dict(out_dim_latent=0, out_dim_position=6))
    mdisc = mk(1, 1, "ADADiscriminator", {})
    tr = Trainer({"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
                  "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
                  "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4),
                  "opt_mask_discriminator": FusedRMSprop(mdisc.parameters(), lr=1e-4)})
    g = torch.Generator().manual_seed(1)
    rows = []
    for it in range(iters):
        data = {"image": torch.rand(1, 3, R, R, generator=g).cuda(), "mask": (torch.rand(1, 1, R, R, generator=g) > 0.5).float().cuda()}
        torch.manual_seed(100 + it)
        np.random.seed(100 + it)
        o = tr.train_step(data)
        rows.append([float(o[k].detach()) if torch.is_tensor(o[k]) else float(o[k]) for k in TRAJ_KEYS])
    return np.array(rows)


def test_training_trajectory_f16x3_tracks_native_fp32():
    """Six full training iterations (G / D / mask-D steps, fused optimisers) from identical seeds in the default
    f16x3 operand mode and in native fp32 MFMA: every logged loss agrees to 1e-4 while the dynamics are still
    deterministic enough to compare (GAN training is chaotic: by iteration ~20 two fp32 runs differ as much)."""
    a, b = _trajectory("f32"), _trajectory("f16x3")
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.abs(a - b).max() < 1e-4, np.abs(a - b).max(axis=1)


# bf16 operand mode (BASELINE configs[1]) over the same six iterations: per logged loss, <= 3x the worst |bf16 - f32| / max(1, |f32|)
# measured on the MI355X (profiles/r6_gradient_margins.txt, rows training_trajectory_bf16_vs_f32)
# The worst iteration's error depends on the draws of the run: two generator draw schemes were measured in round 6 (two launches /
# one launch for latents + jitter).  Worst over the six iterations: generator/loss 2.45e-4 / 7.6e-4-8.7e-4, generator/eikonal 6.5e-4 /
# 6.5e-4, discriminator/loss 2.45e-4 / 1.9e-4-2.3e-4, discriminator/reg 5.8e-5 / 6.9e-5-9.4e-5, mask_discriminator/loss 1.7e-4 /
# 6.9e-5-5.6e-4 (four runs of the final scheme: the split-K sums of the eager discriminator steps are not bit-stable);
# profiles/r6_gradient_margins.txt holds one of them.  The bars are ~3x the largest value seen, rounded.
BF16_TRAJ_TOL = {"generator/loss": 2.6e-3, "generator/eikonal": 2.6e-3, "discriminator/loss": 2.6e-3, "discriminator/reg": 5e-4,
                 "mask_discriminator/loss": 2e-3}


def test_training_trajectory_bf16_tracks_native_fp32():
    """The same six iterations with the generator in the bf16 operand mode against native fp32: every logged loss of every
    iteration inside a MEASURED bar (BF16_TRAJ_TOL).  Together with test_generator_fit_converges_bf16_like_fp32 this is the
    evidence that the mode BASELINE's configs[1] names trains, not only renders (review, round 5: `bf16_mode.training` was a
    speed figure with 1-5 % per-tensor gradient errors and no training-level check).  Reference behaviour replaced: autograd
    through src/models/fields.py:104-122 and stylesdf/volume_renderer.py:50-61 in fp32."""
    a, b = _trajectory("f32"), _trajectory("bf16")
    assert np.isfinite(a).all() and np.isfinite(b).all()
    rel = np.abs(a - b) / np.maximum(1.0, np.abs(a))
    for j, k in enumerate(TRAJ_KEYS):
        record_margin("training_trajectory_bf16_vs_f32", k, float(rel[:, j].max()))
        assert rel[:, j].max() < BF16_TRAJ_TOL[k], (k, rel[:, j])


def _generator_fit(prec, steps=200, R=16):
    """`steps` Adam steps of the GENERATOR ALONE on a fixed objective that needs no discriminator (and therefore no chaos
    argument): silhouette MSE against a target mask one pixel smaller all round than the initial sphere's + 0.1 * eikonal, from sphere_init.  The
    generator draws its own poses / latents / jitter as in training (two elements); both RNGs are re-seeded with the SAME seed
    before every forward, so every step sees the same rays and the objective is a fixed function of the weights.  Returns the
    loss of every step."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_modules import build_generator
    from oi_amd.optim import FusedAdam
    torch.manual_seed(0)
    np.random.seed(0)
    gen = build_generator(R, 16, 16, 1, prec).train()
    bs = 2

    def forward():
        torch.manual_seed(4242)
        np.random.seed(4242)
        return gen(bs=bs, it=0, data={})["box"]

    gen.renderer.pack.set_precision("f32")   # (the target is the same for every mode)
    m0 = forward()["render_out"]["mask"].detach()
    gen.renderer.pack.set_precision(prec)
    # the silhouette to fit: the initial one (28 % of a 16 x 16 crop) eroded by one pixel on every side
    target = (torch.nn.functional.avg_pool2d(m0, 3, stride=1, padding=1) > 0.98).float()
    assert 0.03 < float(target.mean()) < float(m0.mean()) - 0.03, (float(target.mean()), float(m0.mean()))
    opt = FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9))   # the reference's generator optimiser (configs/train.yaml:133-139)
    losses = []
    for it in range(steps):
        opt.zero_grad(set_to_none=False)
        box = forward()
        loss = ((box["render_out"]["mask"] - target) ** 2).mean() + 0.1 * box["loss"]["eikonal"]
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return torch.stack(losses).cpu().numpy()


# measured on the MI355X (tools/dbg/fit_curves.py, three runs per mode: the MLP backward adds its weight gradients with atomics, so
# runs differ in the last bits and Adam with beta1 = 0 amplifies that; profiles/r6_gradient_margins.txt, rows
# generator_fit_200_adam_steps): mean loss of the last 50 steps / first loss 0.019-0.020 in BOTH modes, bf16 / fp32 0.95-1.05, the
# 20-step block means of the two modes within 10 % of each other all the way; the first step's losses agree to 1.2e-5.
# (With lr = 1e-4 the same optimiser oscillates -- last-50 means 0.09-0.14 run to run in either mode --, with betas = (0.9, 0.999) the
# two modes' curves agree to four digits and fall 1000 x: the reference's own setting is the one asserted.)
FIT_DROP = 0.06         # both modes: mean loss of the last 50 steps < FIT_DROP x the first step's (3 x the measured 0.02)
FIT_BF16_VS_F32 = 1.25  # bf16's final loss within this factor of fp32's, either way (measured 0.95-1.05)


def test_generator_fit_converges_bf16_like_fp32():
    """200 steps of the reference's generator optimiser (Adam, lr 2e-5, betas (0, 0.9)) on the generator alone, on a fixed
    silhouette + eikonal objective from sphere_init (`_generator_fit`), in native fp32 and in the bf16 operand mode: both reduce
    the loss by the stated factor and end within the stated factor of each other.  bf16's per-tensor gradient errors (1-5 %,
    section 5 of DESIGN.md) are unbiased enough for Adam to converge to the same place; if this ever fails, bench.py's
    `bf16_mode.training` must stop being quoted."""
    a, b = _generator_fit("f32"), _generator_fit("bf16")
    assert np.isfinite(a).all() and np.isfinite(b).all()
    fa, fb = float(a[-50:].mean()), float(b[-50:].mean())
    record_margin("generator_fit_200_adam_steps", "f32 final / first", fa / float(a[0]))
    record_margin("generator_fit_200_adam_steps", "bf16 final / first", fb / float(b[0]))
    record_margin("generator_fit_200_adam_steps", "bf16 final / f32 final", fb / fa)
    record_margin("generator_fit_200_adam_steps", "first step |bf16 - f32| / f32", abs(float(b[0]) - float(a[0])) / float(a[0]))
    assert fa < FIT_DROP * float(a[0]) and fb < FIT_DROP * float(b[0]), (a[0], fa, b[0], fb)
    assert 1 / FIT_BF16_VS_F32 < fb / fa < FIT_BF16_VS_F32, (fa, fb)


@pytest.mark.parametrize("bs", [1, 2])
def test_graphed_discriminator_steps_match_eager(bs):
    """oi_amd.graphed.GraphedDStep (one hipGraphLaunch per discriminator step, shape-static ADA margins, real and fake batch in
    ONE discriminator pass) against the eager trainer (two passes): three training iterations from identical weights / seeds
    give the same losses and the same weights."""
    import bench
    from oi_amd.config import build_from_config
    from oi_amd.optim import FusedAdam, FusedRMSprop
    from oi_amd.trainer import Trainer
    dev = torch.device("cuda")
    R = 32
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    res = []
    for graphed in (False, True):
        torch.manual_seed(5)
        np.random.seed(5)
        gen, disc = bench.build_models(R, 8, 8, 1, "f16x3", dev)
        mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator",
                                      aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1,
                                      img_size=R, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).to(dev)
        mods = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc,
                "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
                "opt_discriminator": FusedRMSprop(disc.parameters(), lr=1e-4),
                "opt_mask_discriminator": FusedRMSprop(mdisc.parameters(), lr=1e-4)}
        tr = Trainer(mods, graph_d_steps=graphed)
        g = torch.Generator(device=dev).manual_seed(9)
        data = {"image": torch.rand(bs, 3, R, R, device=dev, generator=g), "mask": torch.rand(bs, 1, R, R, device=dev, generator=g)}
        out = None
        for step in range(3):
            torch.manual_seed(100 + step)
            np.random.seed(100 + step)
            out = tr.train_step(data)
        torch.cuda.synchronize()
        r = {k: float(v) for k, v in out.items()}
        r["disc_w"] = float(sum(p.double().abs().sum() for p in disc.parameters()))
        r["mdisc_w"] = float(sum(p.double().abs().sum() for p in mdisc.parameters()))
        res.append(r)
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 2e-4 * max(1.0, abs(res[0][k])), (k, res[0][k], res[1][k])


@pytest.mark.parametrize("n,B", [(700, 1), (300, 2)])
def test_mlp_backward_bounded_scratch_chunks_match_single_launch(sdf_sd, col_sd, n, B, monkeypatch):
    """the backward's working memory is a bound (OI_BWD_SCRATCH_MB), not a function of the problem size.  With
    a cap of ONE workgroup tile (256 points) per batch element the library walks the points in ceil(n / 256) chunks (ragged last chunk,
    per-element offsets); every parameter / FiLM gradient must equal the single-launch result up to the order of the fp32
    atomics."""
    from oi_amd import ops
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    from oi_amd.autograd import sdf_mlp
    g = torch.Generator().manual_seed(n + B)
    pts = (torch.rand(B * n, 3, generator=g) * 2.0 - 1.0).cuda()
    cs, cg, cr = torch.randn(B * n, generator=g).cuda(), 0.1 * torch.randn(B * n, 3, generator=g).cuda(), torch.randn(B * n, 3, generator=g).cuda()
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, "f16x3")
    named = [v for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + list(col_net.parameters())

    def grads():
        wh = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=torch.Generator().manual_seed(3))).cuda().requires_grad_(True)
        _, gamma, beta = pack.film(w=wh)
        sdf, grad, rgb, _ = sdf_mlp(pack, pts, gamma, beta, B, True, True, False)
        loss = (sdf * cs).sum() + (grad * cg).sum() + (rgb * cr).sum()
        return torch.autograd.grad(loss, named + [wh])

    full = grads()
    tile_bytes = ops._l.load().oi_mlp_bwd_scratch_bytes(B, 128)
    assert ops._l.load().oi_mlp_bwd_scratch_bytes_capped(B, n, tile_bytes) == tile_bytes
    assert ops._l.load().oi_mlp_bwd_scratch_bytes_capped(B, n, 1) == tile_bytes          # never below one tile
    monkeypatch.setenv("OI_BWD_SCRATCH_MB", str(tile_bytes / (1 << 20)))
    chunked = grads()
    for a, b in zip(full, chunked):
        assert rel_err(b, a) < 2e-5, rel_err(b, a)


def test_pack_status_reports_non_finite_weights(col_sd):
    """the F16X3 images carry a per-image power-of-two scale taken from the image's own max, so every
    finite weight is representable; what cannot be represented (inf / NaN, e.g. a diverged checkpoint) is flagged in the
    packed header and reported by oi_mlp_pack_status instead of silently producing NaN frames."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, "f16x3")
    pack.check()                                     # healthy weights, including a 1e4 outlier below
    with torch.no_grad():
        lin = [m for m in sdf_net.modules() if type(m).__name__ == "FiLMSiren" and tuple(m.weight.shape) == (128, 128)][2]
        lin.weight[5, 7] = 3.0e4
    pack.check()
    with torch.no_grad():
        lin.weight[5, 7] = float("nan")
    with pytest.raises(RuntimeError, match="non-finite"):
        pack.check()
    with torch.no_grad():
        lin.weight[5, 7] = 0.0
        col_net.rgb_linear.bias[1] = float("inf")
    with pytest.raises(RuntimeError, match="non-finite"):
        pack.check()


WIDE_RANGE_TOL = 2e-5  # measured 5.0e-6 worst over 59 tensors; round 2 accepted 2e-3


def test_mlp_backward_wide_dynamic_range_cotangents(sdf_sd, col_sd):
    """The weight-gradient GEMM scales every operand by ONE power of two per launch (the launch-wide maximum published by
    the sweep).  Cotangents spread over 8 decades between points -- a few surface samples carrying almost all of the loss,
    many far samples carrying very little -- must still give parameter gradients at the fp64 oracle's level."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    from oi_amd.autograd import sdf_mlp
    n, B = 1536, 1
    g = torch.Generator().manual_seed(99)
    pts = torch.rand(B * n, 3, generator=g) * 2.0 - 1.0
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    mag = 10.0 ** (torch.rand(B * n, generator=g) * 8.0 - 6.0)           # 1e-6 .. 1e2 per point
    cs = torch.randn(B * n, generator=g) * mag
    cg = 0.1 * torch.randn(B * n, 3, generator=g) * mag[:, None]
    cr = torch.randn(B * n, 3, generator=g) * mag[:, None]
    loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr)
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, "f16x3")
    wh = w.cuda().requires_grad_(True)
    _, gamma, beta = pack.film(w=wh)
    sdf, grad, rgb, _ = sdf_mlp(pack, pts.cuda(), gamma, beta, B, True, True, False)
    loss = (sdf * cs.cuda()).sum() + (grad * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
    named = [("sdf." + k, v) for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + \
            [("col." + k, v) for k, v in col_net.named_parameters()] + [("w", wh)]
    gr = torch.autograd.grad(loss, [v for _, v in named])
    errs = {name: rel_err(a, g_o[name]) for (name, _), a in zip(named, gr)}
    for name, e in errs.items():
        record_margin("mlp_backward_cotangents_over_8_decades_vs_fp64_oracle[f16x3]", name, e)
    bad = {k: v for k, v in errs.items() if v > WIDE_RANGE_TOL}
    assert not bad, bad


GAMMA_ZERO_TOL = 2e-5


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_mlp_backward_gamma_through_zero(sdf_sd, col_sd, precision):
    """FiLM scales gamma = 15 lin(w) + 30 that cross zero (a trained mapping network can produce lin(w) = -2): rows with
    gamma exactly 0, |gamma| < 1e-2 and negative gamma in every layer kind (layer 0, MFMA layers, colour head), gamma spread
    over about [-20, 80].  The reference (autograd through fields.py:104-122) has no restriction on gamma; the round-3
    backward divided by it.  Every parameter gradient against fp64 oracle autograd, incl. the FiLM heads'."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    from oi_amd.autograd import sdf_mlp
    n, B = 200, 2
    g = torch.Generator().manual_seed(4321)
    sdf_sd = {k: v.clone() for k, v in sdf_sd.items()}
    col_sd = {k: v.clone() for k, v in col_sd.items()}
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))

    def doctor(sd, prefix, rows):
        # spread: gamma - 30 scaled up so that gamma covers about [-20, 80]; then pin chosen rows to exact targets
        sd[prefix + "gamma.weight"] = sd[prefix + "gamma.weight"] * 1.0 + 0.2 * torch.randn(128, 64, generator=g)
        for f, target in rows:
            sd[prefix + "gamma.weight"][f] = 0.0
            sd[prefix + "gamma.bias"][f] = (target - 30.0) / 15.0
    exact = [(3, 0.0), (17, 3e-3), (40, -7e-3), (77, -12.0), (101, 1e-4), (127, 0.0)]
    for l in (0, 1, 4, 7):
        doctor(sdf_sd, f"pts_linears.{l}.", exact)
    doctor(col_sd, "views_linears.", exact)
    pts = torch.rand(B * n, 3, generator=g) * 2.0 - 1.0
    cs, cg, cr = torch.randn(B * n, generator=g), 0.1 * torch.randn(B * n, 3, generator=g), torch.randn(B * n, 3, generator=g)
    loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts, w, cs, cg, cr)

    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW)
    sdf_net.load_state_dict(sdf_sd)
    sdf_net = sdf_net.cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, precision)
    wh = w.cuda().requires_grad_(True)
    _, gamma, beta = pack.film(w=wh)
    gmat = gamma.detach().cpu()
    assert float(gmat.min()) < -10.0 and float(gmat.max()) > 60.0 and int((gmat == 0).sum()) >= 2 * B, \
        (float(gmat.min()), float(gmat.max()), int((gmat == 0).sum()))
    assert int((gmat.abs() < 1e-2).sum()) >= 5 * 4 * B
    sdf, grad, rgb, _ = sdf_mlp(pack, pts.cuda(), gamma, beta, B, True, True, False)
    loss = (sdf * cs.cuda()).sum() + (grad * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
    assert abs(float(loss) - loss_o) < 1e-3 * max(1.0, abs(loss_o))
    named = [("sdf." + k, v) for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + \
            [("col." + k, v) for k, v in col_net.named_parameters()] + [("w", wh)]
    gr = torch.autograd.grad(loss, [v for _, v in named])
    errs = {}
    for (name, _), a in zip(named, gr):
        assert bool(torch.isfinite(a).all()), name
        errs[name] = rel_err(a, g_o[name])
        record_margin(f"mlp_backward_gamma_through_zero_vs_fp64_oracle[{precision}]", name, errs[name])
    bad = {k: v for k, v in errs.items() if v > GAMMA_ZERO_TOL}
    assert not bad, bad
    # the pinned rows themselves (d gamma.bias of a row with gamma = 0 is a sum the round-3 identity could not form)
    for l in (0, 1, 4, 7):
        a = dict(zip([n_ for n_, _ in named], gr))[f"sdf.pts_linears.{l}.gamma.bias"].cpu().double()
        b = g_o[f"sdf.pts_linears.{l}.gamma.bias"]
        rows = [f for f, _ in exact]
        assert float((a[rows] - b[rows]).abs().max()) < 1e-4 * float(b.abs().max()), (l, a[rows], b[rows])


@pytest.mark.parametrize("B", [1, 3])
def test_discriminator_preactivation_chain_matches_layerwise_autograd(B):
    """DCDiscriminator under autograd as a chain of pre-activations (autograd_conv._ConvPre: LeakyReLU applied by the consumer
    on load, one backward launch per layer) against conv + activation per layer: logits, the R1 inner gradient, and every
    parameter gradient of BCE + 10 R1 (first- and second-order paths)."""
    from oi_amd import autograd_conv as AC
    from oi_amd.discriminator import DCDiscriminator
    from oi_amd.losses import gan_losses, grad_wrt_input
    torch.manual_seed(11)
    D = DCDiscriminator(in_dim=3, out_dim=7, n_feat=64, img_size=32, last_bias=True).cuda()
    x0 = torch.randn(B, 3, 32, 32, device="cuda")
    res = []
    for pre in (False, True):
        AC.PRE_CHAIN = pre
        try:
            for p in D.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            d = D(x)
            gx = grad_wrt_input(d[:, :1], x)
            loss, _ = gan_losses(d_real=d, gx=gx, reg_w=10.0)
            loss.backward()
            res.append((d.detach().clone(), gx.detach().clone(), [p.grad.clone() for p in D.parameters()]))
        finally:
            AC.PRE_CHAIN = True
    (d0, g0, p0), (d1, g1, p1) = res
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    assert rel(d1, d0) < 2e-6 and rel(g1, g0) < 2e-6, (rel(d1, d0), rel(g1, g0))
    for a, b in zip(p1, p0):
        assert rel(a, b) < 5e-6, rel(a, b)
