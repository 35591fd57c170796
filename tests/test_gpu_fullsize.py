"""Full-size runs of BASELINE.json's configurations (C2, C4, C5) on the GPU: size-independent properties
(sorted samples, weight normalisation, value ranges, linearity of the compositing in rgb, chunk invariance)
plus oracle agreement on a random subset of rays / points (the oracle cannot run these sizes in seconds)."""
import os

import numpy as np
import pytest
import torch

import oi_oracle as O
from conftest import GOLDEN, load_golden, maxdiff, record_margin

pytestmark = pytest.mark.gpu
NET_KW = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
SDF_NPZ = os.path.join(GOLDEN, "weights_sdf.npz")


def _renderer(col_sd, S, I, K, precision="f32"):
    from oi_amd.fields import ShapeNetwork, ColorNetwork, SingleVarianceNetwork
    from oi_amd.renderer import NeuSRenderer
    sdf = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col = ColorNetwork(**NET_KW)
    col.load_state_dict(col_sd)
    return NeuSRenderer(None, sdf, SingleVarianceNetwork(0.3).cuda(), col.cuda(), n_samples=S, n_importance=I,
                        n_outside=0, up_sample_steps=K, perturb=0, precision=precision)


def _rays(N, seed):
    g = torch.Generator().manual_seed(seed)
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.12 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    return ro, rd, near, far


# per-ray outputs of ALL sampled rays (flipped importance samples included): <= 3x the worst error measured on the MI355X
# (profiles/r6_gradient_margins.txt, rows full_size_all_rays_vs_fp32_oracle)
# measured: colour 4.2e-7 .. 6.4e-7, opacity 8.6e-7 .. 1.27e-6 over the four cases (C4: 0.8 % of the sampled rays unmatched in z,
# and their colour still agrees to 6.3e-7 -- a flipped sample carries almost no weight)
ALL_RAYS_TOL = {1: {"color_fine": 2e-6, "weight_sum": 4e-6}, 4: {"color_fine": 2e-6, "weight_sum": 4e-6}}


@pytest.mark.parametrize("name,B,R,S,I,K,precision", [("C2", 1, 64, 64, 64, 1, "f16x3"), ("C2", 1, 64, 64, 64, 1, "bf16x6"), ("C2", 1, 64, 64, 64, 1, "f32"),
                                                      ("C4", 1, 128, 128, 128, 4, "f16x3")])
def test_full_size_render_properties(sdf_sd, col_sd, name, B, R, S, I, K, precision):
    N = B * R * R
    ro, rd, near, far = _rays(N, 11)
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=torch.Generator().manual_seed(5)))
    r = _renderer(col_sd, S, I, K, precision)
    with torch.no_grad():
        out = r.render(ro.cuda(), rd.cuda(), near.cuda(), far.cuda(), perturb_overwrite=0, cos_anneal_ratio=0.5, w=w.cuda())
        T = S + I
        z = out["mid_z_vals"]
        assert z.shape == (N, T) and bool((z[:, 1:] >= z[:, :-1]).all()), "samples must be sorted along the ray"
        wts = out["weights"]
        assert bool((wts >= 0).all()) and float(out["weight_sum"].max()) <= 1.0 + 1e-4
        assert maxdiff(wts.sum(-1, keepdim=True), out["weight_sum"]) < 1e-5
        assert maxdiff(wts.max(-1, keepdim=True).values, out["weight_max"]) == 0
        rgb = out["raw_color"]
        assert float(rgb.min()) >= 0 and float(rgb.max()) <= 1 and bool(torch.isfinite(out["gradients"]).all())
        assert maxdiff((rgb * wts[..., None]).sum(1), out["color_fine"]) < 2e-5
        # rays through the middle of the (sphere-initialised) object are opaque; eikonal term finite and >= 0
        assert 0.0 <= float(out["gradient_error"]) < 1e3
        hit = (ro + rd * (-(ro * rd).sum(-1, keepdim=True))).norm(dim=-1) < 0.5
        assert float(out["weight_sum"].cpu()[hit].mean()) > 0.9
        # ray independence: rendering a slice of the rays alone gives the same result (chunk invariance)
        sl = slice(N // 3, N // 3 + 256)
        sub = r.render(ro[sl].cuda(), rd[sl].cuda(), near[sl].cuda(), far[sl].cuda(), perturb_overwrite=0,
                       cos_anneal_ratio=0.5, w=w.cuda())
        assert maxdiff(sub["color_fine"], out["color_fine"][sl]) < 1e-6
    # oracle agreement on a subset of rays (K>1: see test_render_vs_oracle_hierarchical for the ray-wise criterion)
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:256]
    ref = O.render(sdf_sd, col_sd, torch.tensor(0.3), ro[idx], rd[idx], near[idx], far[idx], w, S, I, K, 0.5)
    dz = (out["mid_z_vals"].cpu()[idx] - ref["mid_z_vals"]).abs().max(-1).values
    ok = dz < 1e-4
    assert ok.float().mean() >= (0.99 if K == 1 else 0.9)
    assert maxdiff(out["color_fine"].cpu()[idx][ok], ref["color_fine"][ok]) < 2e-4
    # K = 4: four resampling rounds amplify last-bit differences of the coarse sdf values (kernel and oracle are both fp32
    # evaluations; against fp64 the sdf-only kernel is off by 1.5e-6 at most, exactly like the native-fp32 mode --
    # tools/dbg/sdf_err.py); measured 2.1e-4 on the matched rays
    assert maxdiff(out["weights"].cpu()[idx][ok], ref["weights"][ok]) < (2e-4 if K == 1 else 3e-4)
    # ... and EVERY sampled ray, matched or not, is compared where a flipped importance sample cannot hide (review, round 5: at C4
    # one ray in ten may be placed differently and was then not compared at all): the per-ray outputs.  A flip moves one sample of
    # 128 / 256 to a neighbouring section; the composited colour and opacity move by what that sample weighs.
    e_col = maxdiff(out["color_fine"].cpu()[idx], ref["color_fine"])
    e_ws = maxdiff(out["weight_sum"].cpu()[idx], ref["weight_sum"])
    record_margin(f"full_size_all_rays_vs_fp32_oracle[{name}-{precision}]", "color_fine", e_col)
    record_margin(f"full_size_all_rays_vs_fp32_oracle[{name}-{precision}]", "weight_sum", e_ws)
    record_margin(f"full_size_all_rays_vs_fp32_oracle[{name}-{precision}]", "unmatched fraction", 1.0 - float(ok.float().mean()))
    assert e_col < ALL_RAYS_TOL[K]["color_fine"] and e_ws < ALL_RAYS_TOL[K]["weight_sum"], (e_col, e_ws)


@pytest.mark.parametrize("name,B,R,S,I,K", [("C2", 1, 64, 64, 64, 1)])
def test_full_size_render_properties_bf16(sdf_sd, col_sd, name, B, R, S, I, K):
    """BASELINE.json configs[1] in the mode it names (64x64, 64+64 samples/ray, bf16 operands): the size-independent
    properties of every mode, chunk invariance, and the per-ray outputs against the fp32 oracle on 256 rays with the bf16
    mode's stated tolerance (per-sample outputs are not compared: see tests/test_gpu_modules.py, BF16_RAY_TOL)."""
    from conftest import record_margin
    N = B * R * R
    ro, rd, near, far = _rays(N, 11)
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=torch.Generator().manual_seed(5)))
    r = _renderer(col_sd, S, I, K, "bf16")
    with torch.no_grad():
        out = r.render(ro.cuda(), rd.cuda(), near.cuda(), far.cuda(), perturb_overwrite=0, cos_anneal_ratio=0.5, w=w.cuda())
        T = S + I
        z = out["mid_z_vals"]
        assert z.shape == (N, T) and bool((z[:, 1:] >= z[:, :-1]).all()), "samples must be sorted along the ray"
        wts = out["weights"]
        assert bool((wts >= 0).all()) and float(out["weight_sum"].max()) <= 1.0 + 1e-4
        assert maxdiff(wts.sum(-1, keepdim=True), out["weight_sum"]) < 1e-5
        assert maxdiff(wts.max(-1, keepdim=True).values, out["weight_max"]) == 0
        rgb = out["raw_color"]
        assert float(rgb.min()) >= 0 and float(rgb.max()) <= 1 and bool(torch.isfinite(out["gradients"]).all())
        assert maxdiff((rgb * wts[..., None]).sum(1), out["color_fine"]) < 2e-5
        assert 0.0 <= float(out["gradient_error"]) < 1e3
        hit = (ro + rd * (-(ro * rd).sum(-1, keepdim=True))).norm(dim=-1) < 0.5
        assert float(out["weight_sum"].cpu()[hit].mean()) > 0.9
        sl = slice(N // 3, N // 3 + 256)
        sub = r.render(ro[sl].cuda(), rd[sl].cuda(), near[sl].cuda(), far[sl].cuda(), perturb_overwrite=0,
                       cos_anneal_ratio=0.5, w=w.cuda())
        assert maxdiff(sub["color_fine"], out["color_fine"][sl]) < 1e-6   # ray independence holds in every mode
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:256]
    ref = O.render(sdf_sd, col_sd, torch.tensor(0.3), ro[idx], rd[idx], near[idx], far[idx], w, S, I, K, 0.5)
    for k, tol in (("color_fine", 7e-3), ("weight_sum", 1.2e-2)):   # measured 2.1e-3 / 3.9e-3 (means 1.0e-4 / 1.7e-4)
        err = maxdiff(out[k].cpu()[idx], ref[k])
        mae = float((out[k].cpu()[idx] - ref[k]).abs().mean())
        record_margin("c2_full_size_bf16_mode_vs_fp32_oracle", k, err)
        record_margin("c2_full_size_bf16_mode_vs_fp32_oracle", k + "(mean)", mae)
        assert err < tol and mae < 0.1 * tol, (k, err, mae)
    # the normals the shading consumes: the angle between the bf16 mode's and the oracle's weighted normal
    n_a = (out["gradients"].cpu()[idx] * out["weights"].cpu()[idx][..., None]).sum(1)
    n_b = (ref["gradients"] * ref["weights"][..., None]).sum(1)
    solid = ref["weight_sum"].squeeze(-1) > 0.5
    cosang = torch.nn.functional.cosine_similarity(n_a[solid], n_b[solid], dim=-1)
    record_margin("c2_full_size_bf16_mode_vs_fp32_oracle", "normal angle (rad)", float(torch.acos(cosang.clamp(-1, 1)).max()))
    assert float(cosang.min()) > 0.992, float(cosang.min())   # measured: 4.2e-2 rad worst; 0.992 = 0.127 rad


def test_c5_mlp_only_microbench_size(sdf_sd, col_sd):
    """C5: 2^20 rays x 512 points through the SDF network (sdf-only kernel, chunked over rays); oracle on a
    random subsample; determinism across two runs; linear ray parametrisation consistency."""
    from oi_amd.autograd import sdf_mlp
    r = _renderer(col_sd, 64, 64, 1)
    w = O.style_mlp(sdf_sd, torch.randn(1, 64, generator=torch.Generator().manual_seed(9)))
    _, gamma, beta = r.pack.film(w=w.cuda())
    n_rays, n_samp, chunk = 1 << 20, 512, 1 << 15
    g = torch.Generator(device="cuda").manual_seed(0)
    checks = []
    with torch.no_grad():
        for c0 in range(0, n_rays, chunk):
            d = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0], device="cuda") +
                                              0.1 * torch.randn(chunk, 3, device="cuda", generator=g), dim=-1)
            t = 2.0 + 2.0 * torch.rand(chunk, n_samp, 1, device="cuda", generator=g)
            pts = (torch.tensor([0.0, 0.0, -3.0], device="cuda") + d[:, None, :] * t).reshape(-1, 3)
            sdf = sdf_mlp(r.pack, pts, gamma, beta, 1, False, False, False)[0]
            if c0 % (chunk * 8) == 0:
                sel = torch.randint(0, pts.shape[0], (64,), device="cuda", generator=g)
                checks.append((pts[sel].cpu(), sdf[sel].cpu()))
                sdf2 = sdf_mlp(r.pack, pts, gamma, beta, 1, False, False, False)[0]
                assert torch.equal(sdf, sdf2), "kernel must be deterministic"
    assert torch.isfinite(sdf).all()
    p = torch.cat([c[0] for c in checks])
    s = torch.cat([c[1] for c in checks])
    ref = O.sdf_forward(sdf_sd, p, w)[0].squeeze(-1)
    assert maxdiff(s, ref) < 2e-5


# 3x the error measured in the native-fp32 mode against fp64 autograd through the oracle: 8.8e-6 (MLP op, 59 tensors),
# 6.4e-6 (training render, 58 tensors) -- tools/grad_margin.py; round 2 accepted 2e-3 / 3e-3
C2_MLP_BWD_TOL = 2.5e-5
C2_RENDER_BWD_TOL = 2e-5
# bf16 operand mode (BASELINE configs[1]): <= 3x the worst error measured on the MI355X at the FINAL kernels
# (profiles/r6_gradient_margins.txt, rows c2_size_*[bf16]; re-measured in round 6 -- the round-5 bars 0.12 / 0.32 dated from the
# round-4 forward kernel and sat 20x above the committed measurement of the render row):
#   MLP op 3.83e-2 worst / 1.36e-2 median over 60 tensors, loss 8.1e-5;
#   training render 1.56e-2 worst / 6.4e-3 median over 59, loss 3.2e-3 (ONE scalar: 512 randomly weighted colour / weight-sum
#   terms of 128 rays; the compositing turns sdf errors into weight errors at 1/s_val).
C2_MLP_BWD_TOL_BF16 = 0.115
C2_RENDER_BWD_TOL_BF16 = 0.047
C2_LOSS_TOL_BF16 = 1e-2


@pytest.mark.parametrize("n,precision", [(4096 * 128, "f16x3"), (4096 * 128, "f32"), (128 * 128 * 256, "f16x3"), (4096 * 128, "bf16")],
                         ids=["C2-f16x3", "C2-f32", "C4-eight-chunks-f16x3", "C2-bf16"])
def test_c2_size_mlp_backward_vs_oracle(sdf_sd, col_sd, n, precision):
    """Backward of the MLP op at the full C2 launch size (524,288 points = 4,096 rays x 128 samples; the real grid of the
    sweep kernel, its scratch indexing, the split weight-gradient GEMM and every atomics-accumulated output): the
    cotangents are non-zero on 2,048 points scattered over the launch, so the parameter gradients equal those of the
    subset alone, which the fp64 oracle differentiates with autograd (points are independent in this op)."""
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    from oi_amd.autograd import sdf_mlp
    from test_gpu_backward import _oracle_mlp_grads, rel_err
    # (C4 per rank: 128 x 128 rays x 256 samples = 4,194,304 points -- the backward then walks the points in eight chunks
    #  of the default 9 GiB working-memory bound, every chunk accumulating into the same outputs)
    B, n_sub = 1, 2048
    g = torch.Generator().manual_seed(42)
    pts = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    idx = torch.randperm(n, generator=g)[:n_sub]
    cs, cg, cr = torch.zeros(n), torch.zeros(n, 3), torch.zeros(n, 3)
    cs[idx] = torch.randn(n_sub, generator=g)
    cg[idx] = 0.1 * torch.randn(n_sub, 3, generator=g)
    cr[idx] = torch.randn(n_sub, 3, generator=g)
    loss_o, g_o = _oracle_mlp_grads(sdf_sd, col_sd, pts[idx], w, cs[idx], cg[idx], cr[idx])
    sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda()
    col_net = ColorNetwork(**NET_KW)
    col_net.load_state_dict(col_sd)
    col_net = col_net.cuda()
    pack = FieldPack(sdf_net, col_net, precision)
    wh = w.cuda().requires_grad_(True)
    _, gamma, beta = pack.film(w=wh)
    sdf, grad, rgb, _ = sdf_mlp(pack, pts.cuda(), gamma, beta, B, True, True, False)
    loss = (sdf * cs.cuda()).sum() + (grad * cg.cuda()).sum() + (rgb * cr.cuda()).sum()
    bf = precision == "bf16"
    case = f"{'c2' if n < 1 << 20 else 'c4'}_size_mlp_backward_vs_fp64_oracle[{precision}]"
    if bf:
        record_margin(case, "(loss)", abs(float(loss) - loss_o) / max(1.0, abs(loss_o)))
    assert abs(float(loss) - loss_o) < (C2_LOSS_TOL_BF16 if bf else 1e-3) * max(1.0, abs(loss_o))
    named = [("sdf." + k, v) for k, v in sdf_net.named_parameters() if not k.startswith("style.")] + \
            [("col." + k, v) for k, v in col_net.named_parameters()] + [("w", wh)]
    gr = torch.autograd.grad(loss, [v for _, v in named])
    errs = {name: rel_err(a, g_o[name]) for (name, _), a in zip(named, gr)}
    for name, e in errs.items():
        record_margin(case, name, e)
    bad = {k: v for k, v in errs.items() if v > (C2_MLP_BWD_TOL_BF16 if bf else C2_MLP_BWD_TOL)}
    assert not bad, bad


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16"])
def test_c2_size_render_backward_ray_subset_vs_oracle(sdf_sd, col_sd, precision):
    """Training render at C2 size (4,096 rays x 64+64 samples, gradient enabled): loss = <c, color_fine> + <c', weight_sum>
    with cotangents on a 128-ray subset.  Rays are independent, so the parameter gradients equal those of the subset
    alone, which the fp64 oracle differentiates with autograd (render_core on the HIP path's own samples of those rays:
    the sampling itself carries no gradient, renderer.py:390).  The global eikonal mean is pinned by F6 / F9 and
    tests/test_gpu_backward.py::test_composite_backward_vs_oracle."""
    from test_gpu_backward import rel_err
    N, S, I = 4096, 64, 64
    ro, rd, near, far = _rays(N, 23)
    w = O.style_mlp(sdf_sd, torch.randn(1, 64, generator=torch.Generator().manual_seed(6)))
    r = _renderer(col_sd, S, I, 1, precision)
    g = torch.Generator().manual_seed(7)
    idx = torch.sort(torch.randperm(N, generator=g)[:128]).values
    c_col, c_ws = torch.zeros(N, 3), torch.zeros(N, 1)
    c_col[idx] = torch.randn(128, 3, generator=g)
    c_ws[idx] = torch.randn(128, 1, generator=g)
    named = [("sdf." + k, v) for k, v in r.sdf_network.named_parameters()] + \
            [("col." + k, v) for k, v in r.color_network.named_parameters()]
    smp, comp = r.render_full(ro.cuda(), rd.cuda(), near.cuda(), far.cuda(), 0, 0.5, None, w.cuda(),
                              outputs=("weights", "weight_sum", "color_fine", "reduce4"))
    loss = (comp["color_fine"] * c_col.cuda()).sum() + (comp["weight_sum"] * c_ws.cuda()).sum()
    gr = torch.autograd.grad(loss, [v for _, v in named], allow_unused=True)
    sd = {k: v.double().clone().requires_grad_(True) for k, v in sdf_sd.items()}
    csd = {k: v.double().clone().requires_grad_(True) for k, v in col_sd.items()}
    z = smp["z_vals"].detach().cpu()[idx].double()
    ref = O.render_core(sd, csd, torch.tensor(0.3, dtype=torch.float64), ro[idx].double(), rd[idx].double(), z,
                        w.double(), S, 0.5)
    loss_o = (ref["color_fine"] * c_col[idx].double()).sum() + (ref["weight_sum"] * c_ws[idx].double()).sum()
    bf = precision == "bf16"
    if bf:
        record_margin("c2_size_render_backward_vs_fp64_oracle[bf16]", "(loss)", abs(float(loss) - float(loss_o)) / max(1.0, abs(float(loss_o))))
    assert abs(float(loss) - float(loss_o)) < (C2_LOSS_TOL_BF16 if bf else 1e-3) * max(1.0, abs(float(loss_o)))
    leaves = [("sdf." + k, v) for k, v in sd.items()] + [("col." + k, v) for k, v in csd.items()]
    g_o = dict(zip([n_ for n_, _ in leaves], torch.autograd.grad(loss_o, [v for _, v in leaves], allow_unused=True)))
    bad, checked = {}, 0
    for (name, _), a in zip(named, gr):
        b = g_o.get(name)
        if a is None or b is None or name.startswith("sdf.style."):
            continue
        checked += 1
        record_margin(f"c2_size_render_backward_vs_fp64_oracle[{precision}]", name, rel_err(a, b))
        if rel_err(a, b) > (C2_RENDER_BWD_TOL_BF16 if bf else C2_RENDER_BWD_TOL):
            bad[name] = rel_err(a, b)
    assert checked > 50 and not bad, bad
