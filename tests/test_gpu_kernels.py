"""GPU parity tests: every HIP kernel of liboi_hip.so (called through the C ABI via oi_amd.ops)
against the oracle on the same seeded inputs and against the committed golden fixtures.
Tolerances: fp32 / bf16x3 paths 1e-4 (north_star); bf16 path stated per test."""
import math

import numpy as np
import pytest
import torch

import oi_oracle as O
O_ref = O
from conftest import load_golden, maxdiff, sub_sd, record_margin

pytestmark = pytest.mark.gpu


def dev(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.fixture(scope="module")
def ops():
    from oi_amd import ops as _ops, lib
    lib.load()
    return _ops


@pytest.fixture(scope="module")
def packed_all(ops, sdf_sd, col_sd):
    from oi_amd.params import stack_field_params
    P = stack_field_params(dev(sdf_sd), dev(col_sd))
    packs = {}
    for name, prec in (("f32", 0), ("bf16x3", 1), ("bf16", 2), ("bf16x6", 3), ("f16x3", 4)):
        packs[name] = ops.mlp_pack_weights(P["w0"], P["b0"], P["wh"], P["bh"], P["wsig"], P["bsig"], P["wv"], P["bv"],
                                           P["wrgb"], P["brgb"], prec)
    return P, packs


def test_lib_identity(ops):
    from oi_amd import lib
    L = lib.load()
    assert L.oi_arch() == b"gfx950" and L.oi_version() >= 1


def test_film_params(ops, packed_all, sdf_sd, col_sd):
    P, _ = packed_all
    g1 = load_golden("f1_film_siren")
    w, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], z=g1["z"].cuda())
    assert maxdiff(w.cpu(), g1["w"]) < 1e-5
    for l in range(8):
        g, b = O.film_params(sdf_sd, f"pts_linears.{l}.", g1["w"])
        assert maxdiff(gamma[:, l].cpu(), g) < 1e-4 and maxdiff(beta[:, l].cpu(), b) < 1e-5
    g, b = O.film_params(col_sd, "views_linears.", g1["w"])
    assert maxdiff(gamma[:, 8].cpu(), g) < 1e-4 and maxdiff(beta[:, 8].cpu(), b) < 1e-5
    w2, gamma2, _ = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], w=g1["w"].cuda())
    assert maxdiff(gamma2, gamma) < 1e-4


@pytest.mark.parametrize("mode,tol_sdf,tol_grad,tol_rgb", [("f32", 2e-5, 1e-4, 2e-5), ("bf16x6", 2e-5, 1e-4, 2e-5), ("f16x3", 2e-5, 1e-4, 2e-5),
                                                           ("bf16x3", 5e-5, 2e-4, 5e-5), ("bf16", 3e-2, 1.5e-1, 3e-2)])
def test_sdf_mlp_golden(ops, packed_all, col_sd, mode, tol_sdf, tol_grad, tol_rgb):
    """F1/F2: sdf, features, analytic gradient and colour head vs the reference's own outputs."""
    P, packs = packed_all
    g1, g2 = load_golden("f1_film_siren"), load_golden("f2_color")
    prec = {"f32": 0, "bf16x3": 1, "bf16": 2, "bf16x6": 3, "f16x3": 4}[mode]
    w, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], w=g1["w"].cuda())
    sdf, grad, rgb, feat, _ = ops.sdf_mlp_fwd(g1["pts"].cuda(), packs[mode], gamma, beta, 2, prec,
                                              fast_trig=(mode == "bf16"), want_grad=True, want_rgb=True, want_feat=True)
    torch.cuda.synchronize()
    e_sdf, e_feat = maxdiff(sdf.cpu(), g1["sdf"].squeeze(-1)), maxdiff(feat.cpu(), g1["feat"])
    gscale = max(1.0, float(g1["grad"].abs().max()))  # |d sdf/dx| reaches O(10) for random latents
    e_grad, e_rgb = maxdiff(grad.cpu(), g1["grad"]) / gscale, maxdiff(rgb.cpu(), g2["rgb"])
    print(f"[{mode}] sdf {e_sdf:.2e} feat {e_feat:.2e} grad(rel {gscale:.1f}) {e_grad:.2e} rgb {e_rgb:.2e}")
    assert e_sdf < tol_sdf and e_grad < tol_grad and e_rgb < tol_rgb
    # intermediate 128-d features (not a renderer output): bf16x3 drops the lo*lo product terms
    # (2^-16 relative per product, amplified by the gamma ~ 30 FiLM phases over 8 layers)
    assert e_feat < {"f32": 1e-4, "bf16x6": 1e-4, "f16x3": 1e-4, "bf16x3": 3e-4, "bf16": 1e-1}[mode]
    # sdf-only variant agrees with the full variant (f16x3: two kernels, the register-resident one forms the FiLM phase
    # in revolutions: two valid fp32 roundings of the same phase, both within tol_sdf of the reference)
    sdf2, _, _, _, _ = ops.sdf_mlp_fwd(g1["pts"].cuda(), packs[mode], gamma, beta, 2, prec, fast_trig=(mode == "bf16"))
    # (bf16: likewise two kernels -- mlp_fwd3b.hip forms the phase in revolutions and feeds it unreduced to v_sin --, both
    # within the mode's tolerance of the reference)
    assert maxdiff(sdf2, sdf) < {"f16x3": 5e-6, "bf16": tol_sdf}.get(mode, 1e-6)


def test_color_network_forward_standalone_golden_f2(col_sd):
    """ColorNetwork.forward(points, normals, view_dirs, feature_vectors, z, w) on the REFERENCE's own feature vectors and
    normals (F2: fields.py:89-101 evaluated by the reference): the stand-alone entry oi_color_head_fwd."""
    from oi_amd.fields import ColorNetwork
    g2 = load_golden("f2_color")
    col = ColorNetwork(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
    col.load_state_dict(col_sd)
    col = col.cuda()
    n = g2["feat"].shape[0]
    pts = torch.zeros(n, 3, device="cuda")
    with torch.no_grad():
        rgb = col(pts, g2["grad"].cuda(), None, g2["feat"].cuda(), None, g2["w"].cuda())
    assert rgb.shape == (n, 3) and maxdiff(rgb.cpu(), g2["rgb"]) < 2e-5, maxdiff(rgb.cpu(), g2["rgb"])
    # ragged point counts (tile tails), three batch elements, against the oracle
    for npe in (1, 37, 130, 1000):
        gen = torch.Generator().manual_seed(npe)
        B = 3
        feat = torch.rand(B * npe, 128, generator=gen) * 2 - 1
        nrm = torch.randn(B * npe, 3, generator=gen) * 3
        w = torch.randn(B, 64, generator=gen)
        ref = O.color_head(col_sd, feat, nrm, w)
        with torch.no_grad():
            out = col(torch.zeros(B * npe, 3, device="cuda"), nrm.cuda(), None, feat.cuda(), None, w.cuda())
        assert maxdiff(out.cpu(), ref) < 2e-5, (npe, maxdiff(out.cpu(), ref))
    with pytest.raises(ValueError):
        col(pts, g2["grad"].cuda(), None, g2["feat"].cuda()[:-1], None, g2["w"].cuda())


RAGGED_TOL = {"f32": (2e-5, 1e-4, 2e-5), "f16x3": (2e-5, 1e-4, 2e-5), "bf16": (3e-2, 1.5e-1, 3e-2)}


@pytest.mark.parametrize("mode", ["f32", "f16x3", "bf16"])
@pytest.mark.parametrize("n", [1, 37, 128, 1000])
def test_sdf_mlp_ragged_tiles(ops, packed_all, sdf_sd, col_sd, n, mode):
    """Tile tails: n points per element not a multiple of the workgroup tile -- the v2 kernel (f32: 256-point tiles), the
    register-resident f16x3 kernel and the register-resident bf16 kernel (128-point workgroups, CU-indexed slots) each."""
    P, packs = packed_all
    g = torch.Generator().manual_seed(n)
    B = 3
    pts = torch.rand(B * n, 3, generator=g) * 2.4 - 1.2
    z = torch.randn(B, 64, generator=g)
    w = O.style_mlp(sdf_sd, z)
    sdf_o, feat_o, grad_o = O.sdf_forward(sdf_sd, pts, w, want_grad=True)
    rgb_o = O.color_head(col_sd, feat_o, grad_o, w)
    _, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], w=w.cuda())
    prec = {"f32": 0, "bf16": 2, "f16x3": 4}[mode]
    # guard bands around every output: a tail lane that stored past its element would land here
    sdf, grad, rgb, feat, _ = ops.sdf_mlp_fwd(pts.cuda(), packs[mode], gamma, beta, B, prec, fast_trig=(mode == "bf16"),
                                              want_grad=True, want_rgb=True, want_feat=True)
    t_sdf, t_grad, t_rgb = RAGGED_TOL[mode]
    assert bool(torch.isfinite(sdf).all() and torch.isfinite(grad).all() and torch.isfinite(rgb).all())
    assert maxdiff(sdf.cpu(), sdf_o.squeeze(-1)) < t_sdf
    assert maxdiff(grad.cpu(), grad_o) < t_grad * max(1.0, float(grad_o.abs().max()))
    assert maxdiff(rgb.cpu(), rgb_o) < t_rgb
    assert maxdiff(feat.cpu(), feat_o) < {"f32": 1e-4, "f16x3": 1e-4, "bf16": 1e-1}[mode]
    # the sdf-only pass on the same ragged shape
    sdf2 = ops.sdf_mlp_fwd(pts.cuda(), packs[mode], gamma, beta, B, prec, fast_trig=(mode == "bf16"))[0]
    assert maxdiff(sdf2.cpu(), sdf_o.squeeze(-1)) < t_sdf


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
def test_persistent_workgroups_bit_identical(mode, tmp_path):
    """The persistent-workgroup launches (a workgroup walks several tiles: the bf16 full kernel, the sdf-only passes of every
    mode) against one workgroup per tile of the SAME kernels (OI_B3P_PERSIST=0 / OI_V2_PERSIST=0, read once per process: two
    subprocesses): a seeded forward of 2 x 100,003 points -- 782 tiles of 128 (391 of 256) per element on at most 128 workgroups,
    ragged last tile -- must agree bit for bit in sdf, gradient, albedo and features."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for v in ("1", "0"):
        out = str(tmp_path / f"fwd_{mode}_{v}.pt")
        env = dict(os.environ, OI_B3P_PERSIST=v, OI_V2_PERSIST=v)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "fwd_dump.py"), out, mode], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:]
        outs.append(torch.load(out))
    a, b = outs
    assert set(a) == set(b) and len(a) >= 5
    for k in a:
        assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k


def test_q24_slot_format_round_trip():
    """The 24-bit fixed-point slots of the f16x3 backward (csrc/mlp_bwd.hip pack_q24 / unpack_q24, through oi_selftest_q24).
    Values: a lane's 64 entries share one power-of-two scale chosen from their maximum -- the reconstruction error is at most
    2^-21 of that maximum (the resolution of the weight-gradient GEMM's fp16 hi + lo split) whatever the entries' own size, zeros
    and lane maxima included.  Phases: 2^-23 revolutions, and a phase within 2^-24 of 1 -- which rounds to the top of the binade --
    must come back as the SAME ANGLE (1.0 = 0 revolutions), not half a revolution off (the bug the C2-size gradient test caught)."""
    import ctypes
    from oi_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(5)
    # 512 lanes: magnitudes over 30 decades between lanes, 6 decades inside a lane, exact zeros, a lane of zeros, a lane of 1 value
    mag = 10.0 ** (torch.rand(512, 1, generator=g) * 30 - 20)
    x = torch.randn(512, 64, generator=g) * mag * 10.0 ** (-6 * torch.rand(512, 64, generator=g))
    x[::7, 3] = 0.0
    x[5] = 0.0
    x[6] = 0.0
    x[6, 17] = -3.25e-9
    x[9, 0] = x[9].abs().max() * 1.999999   # an entry at the very top of the lane's binade
    x = x.float().cuda().contiguous()
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.oi_selftest_q24(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), x.numel(), 0, ctypes.c_void_p(stream)) == 0
    mx = x.abs().amax(dim=1, keepdim=True)
    err = (y.double() - x.double()).abs()
    assert bool((err <= mx.double() * 2.0 ** -21 + 1e-45).all()), float((err / mx.clamp_min(1e-30)).max())
    assert bool((y[5] == 0).all())
    # phases: a dense sample, the ends of the interval and the values that round to y = 2.0
    r = torch.rand(64 * 256, generator=g)
    r[:8] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 1.0 - 2.0 ** -25, 1.0 - 2.0 ** -23, 0.5, 0.25, 2.0 ** -30, 0.75])
    r = r.float().clamp(max=float(torch.nextafter(torch.tensor(1.0), torch.tensor(0.0)))).cuda().contiguous()
    d = torch.empty_like(r)
    assert L.oi_selftest_q24(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(d.data_ptr()), r.numel(), 1, ctypes.c_void_p(stream)) == 0
    assert bool(((d >= 1.0) & (d <= 2.0)).all())
    ang = 2 * math.pi * r.double()
    dec = 2 * math.pi * (d.double() - 1.0)
    assert float((torch.sin(dec) - torch.sin(ang)).abs().max()) < 1e-6
    assert float((torch.cos(dec) - torch.cos(ang)).abs().max()) < 1e-6


def test_gen_rays(ops):
    g = load_golden("f5_generator")
    R, SR = int(g["resolution"]), int(g["scene_resolution"])
    K, K_inv, c2w, w2c = O.camera_matrices(float(g["cam_dist"]), float(g["scene_fov"]), SR)
    b2w = g["b2w"]
    b2c = w2c @ b2w
    t = b2c[:, :3, 3]
    cd = float(g["cam_dist"])
    offs = torch.stack([cd / t[:, 2] * t[:, 0] * R / 2 + 0.5 * SR - R / 2, cd / t[:, 2] * t[:, 1] * R / 2 + 0.5 * SR - R / 2], -1)
    ro, rd, near, far = ops.gen_rays(g["c2b"].cuda(), K_inv[:3, :3].contiguous().cuda(), offs.cuda(), R)
    assert maxdiff(ro.cpu(), g["rays_o"]) < 1e-5 and maxdiff(rd.cpu(), g["rays_d"]) < 2e-6
    n_o, f_o = O.near_far_from_sphere(g["rays_o"].reshape(-1, 3), g["rays_d"].reshape(-1, 3))
    assert maxdiff(near.cpu(), n_o) < 1e-5 and maxdiff(far.cpu(), f_o) < 1e-5


@pytest.mark.parametrize("S", [16, 64, 7])
def test_coarse_samples_and_midpoints(ops, S):
    g = load_golden("f3_upsample")
    ro, rd, near, far = (g[k] for k in ("rays_o", "rays_d", "near", "far"))
    jit = torch.rand(ro.shape[0], 1, generator=torch.Generator().manual_seed(S))
    for j in (None, jit):
        z_o = O.coarse_z(near, far, S, j)
        z, pts = ops.coarse_samples(ro.cuda(), rd.cuda(), near.cuda(), far.cuda(), S, None if j is None else j.cuda())
        assert maxdiff(z.cpu(), z_o) < 1e-6
        assert maxdiff(pts.cpu(), ro[:, None] + rd[:, None] * z_o[..., None]) < 1e-6
    dists, mid, pts = ops.midpoints(ro.cuda(), rd.cuda(), z, 2.0 / S)
    d_o = torch.cat([z_o[:, 1:] - z_o[:, :-1], torch.full_like(z_o[:, :1], 2.0 / S)], -1)
    assert maxdiff(dists.cpu(), d_o) < 1e-6 and maxdiff(mid.cpu(), z_o + d_o * 0.5) < 1e-6


def test_upsample_golden(ops):
    """F3: up_sample + sample_pdf + merge against the reference's outputs (K=1)."""
    g = load_golden("f3_upsample")
    ro, rd = g["rays_o"].cuda(), g["rays_d"].cuda()
    z_new, pts_new, z_m = ops.upsample(ro, rd, g["z_coarse"].cuda(), g["sdf_coarse"].cuda(), 16, 64.0)
    assert maxdiff(z_new.cpu(), g["z_new_k1"]) < 2e-5
    assert maxdiff(z_m.cpu(), g["z_merged_k1"]) < 2e-5
    assert (z_m[:, 1:] >= z_m[:, :-1]).all()
    assert maxdiff(pts_new.cpu(), g["rays_o"][:, None] + g["rays_d"][:, None] * z_new.cpu()[..., None]) < 1e-6


@pytest.mark.parametrize("Sc,n_new,inv_s", [(16, 4, 64.0), (64, 64, 64.0), (160, 32, 256.0), (128, 128, 512.0), (9, 70, 64.0)])
def test_upsample_vs_oracle(ops, sdf_sd, Sc, n_new, inv_s):
    g = torch.Generator().manual_seed(Sc + n_new)
    N = 203
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.2 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    z = torch.sort(near + (far - near) * torch.rand(N, Sc, generator=g), -1).values
    w = O.style_mlp(sdf_sd, torch.randn(1, 64, generator=g))
    pts = ro[:, None] + rd[:, None] * z[..., None]
    sdf = O.sdf_forward(sdf_sd, pts.reshape(-1, 3), w)[0].reshape(N, Sc)
    wts = O.up_sample_weights(ro, rd, z, sdf, inv_s)
    zn_o = O.sample_pdf_det(z, wts, n_new)
    zm_o, _ = O.merge_sorted(z, zn_o)
    z_new, _, z_m = ops.upsample(ro.cuda(), rd.cuda(), z.cuda(), sdf.cuda(), n_new, inv_s)
    assert maxdiff(z_new.cpu(), zn_o) < 5e-5, maxdiff(z_new.cpu(), zn_o)
    assert maxdiff(z_m.cpu(), zm_o) < 5e-5
    # merge with payload
    sdf_new = torch.randn(N, n_new, generator=g)
    zo, so = ops.merge_sorted(z.cuda(), sdf.cuda(), z_new, sdf_new.cuda())
    zr, sr = O.merge_sorted(z, z_new.cpu(), sdf, sdf_new)
    assert maxdiff(zo.cpu(), zr) == 0.0
    # payload follows its key: the (key, payload) pairs agree as multisets per ray (torch.sort is not
    # stable, so payloads of *equal* keys may come out in a different order)
    a = np.stack([zo.cpu().numpy(), so.cpu().numpy()], -1)
    b = np.stack([zr.numpy(), sr.numpy()], -1)
    for i in range(N):
        ia, ib = np.lexsort((a[i, :, 1], a[i, :, 0])), np.lexsort((b[i, :, 1], b[i, :, 0]))
        assert np.array_equal(a[i][ia], b[i][ib]), i


def _composite_inputs(tag, g):
    return dict(sdf=g[f"{tag}_sdf"], grad=g[f"{tag}_gradients"], rgb=g[f"{tag}_raw_color"], mid_z=g[f"{tag}_mid_z_vals"])


@pytest.mark.parametrize("tag,car", [("c0p0", 0.0), ("c0p5", 0.5), ("c1p0", 1.0)])
def test_composite_golden_f4(ops, tag, car):
    """F4: compositing tail of render_core on the reference's own per-sample network outputs."""
    g = load_golden("f4_render")
    ci = _composite_inputs(tag, g)
    N, T = ci["sdf"].shape
    z = g[f"{tag}_mid_z_vals"]
    # dists are recovered from mid_z: mid = z + d/2 with the last d = 2/S
    S = 16
    dists = torch.empty_like(z)
    dists[:, -1] = 2.0 / S
    zz = torch.empty_like(z)
    zz[:, -1] = z[:, -1] - dists[:, -1] * 0.5
    for i in range(T - 2, -1, -1):
        zz[:, i] = 2 * z[:, i] - zz[:, i + 1]
        dists[:, i] = zz[:, i + 1] - zz[:, i]
    out = ops.composite_fwd(ci["sdf"].cuda(), ci["grad"].cuda(), ci["rgb"].cuda(), dists.cuda(), ci["mid_z"].cuda(),
                            g["rays_o"].cuda(), g["rays_d"].cuda(), torch.tensor([[0.0, 0.0, -1.0]]).cuda(), None,
                            g["variance"].cuda(), torch.tensor([-0.7, 0.0, 10.0]).cuda(), car, 1)
    torch.cuda.synchronize()
    tol = 3e-4  # the recovered dists carry ~1e-6 error amplified by inv_s
    assert maxdiff(out["weights"].cpu(), g[f"{tag}_weights"]) < tol
    assert maxdiff(out["cdf"].cpu(), g[f"{tag}_cdf_fine"]) < tol
    assert maxdiff(out["weight_sum"].cpu(), g[f"{tag}_weight_sum"]) < tol
    assert maxdiff(out["weight_max"].cpu(), g[f"{tag}_weight_max"]) < tol
    assert maxdiff(out["color_fine"].cpu(), g[f"{tag}_color_fine"]) < tol
    assert maxdiff(out["inside_sphere"].cpu(), g[f"{tag}_inside_sphere"]) == 0
    assert maxdiff(out["pts_norm"].cpu(), g[f"{tag}_pts_norm"]) < 1e-5
    r4 = out["reduce4"].cpu()
    assert abs(float(r4[0] / (r4[1] + 1e-5)) - float(g[f"{tag}_gradient_error"])) < 1e-4
    assert abs(float(r4[2] / (N * T)) - float(g[f"{tag}_surface_loss"])) < 1e-5


def test_composite_vs_oracle_maps(ops, sdf_sd, col_sd):
    """Composite + Phong maps (specular on, 2 elements, T not a multiple of 64) vs oracle render_maps."""
    g = torch.Generator().manual_seed(3)
    B, H, W, S, I = 2, 5, 7, 40, 37
    N = B * H * W
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.2 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    w = O.style_mlp(sdf_sd, torch.randn(B, 64, generator=g))
    var = torch.tensor(0.3)
    out_o = O.render(sdf_sd, col_sd, var, ro, rd, near, far, w, S, I, 1, 0.37)
    lsd = {"param_direction": torch.tensor([0.3, -0.5, -0.8]), "param_ambient": torch.tensor(-0.4),
           "param_specular": torch.tensor(0.35), "param_shininess": torch.tensor(6.0)}
    w2b = torch.eye(4).repeat(B, 1, 1)
    q = torch.linalg.qr(torch.randn(B, 3, 3, generator=g)).Q
    w2b[:, :3, :3] = q
    bg = torch.rand(B, 3, generator=g)
    maps = O.render_maps(out_o, ro, lsd, w2b, bg, B, H, W, return_raw=True)
    ldir, amb, cd, cs, sh = O.light_terms(lsd, w2b)
    T = S + I
    dists = torch.cat([out_o["mid_z_vals"][:, 1:] * 0, torch.zeros(N, 1)], -1)
    # exact dists from the oracle's z: recompute
    z = O.hierarchical_z(sdf_sd, ro, rd, near, far, w, S, I, 1)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), 2.0 / S)], -1)
    out = ops.composite_fwd(out_o["sdf"].cuda(), out_o["gradients"].cuda(), out_o["raw_color"].cuda(), dists.cuda(),
                            out_o["mid_z_vals"].cuda(), ro.cuda(), rd.cuda(), ldir.cuda(), bg.cuda(), var.cuda(),
                            torch.stack([lsd["param_ambient"], lsd["param_specular"], lsd["param_shininess"]]).cuda(),
                            0.37, B)
    torch.cuda.synchronize()

    def as_map(x):
        return x.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)

    assert maxdiff(out["weights"].cpu(), out_o["weights"]) < 1e-5
    assert maxdiff(out["alpha"].cpu(), out_o["alpha"]) < 1e-5
    for name, key in (("image", "image"), ("image_no_bg", "image_no_bg"), ("mask", "mask"), ("normal", "normal_map"),
                      ("z_map", "z_map"), ("color_fine", "color_map"), ("weight_sum", "weight_sum_map")):
        assert maxdiff(as_map(out[name]), maps[key]) < 2e-5, name
    assert maxdiff(as_map(out["shading"]), maps["shading_map"][:, :1]) < 2e-5
    assert maxdiff(as_map(out["specular_map"]), maps["specular_map"][:, :1]) < 2e-5
    assert maxdiff(as_map(out["diffuse_map"]), maps["diff_shading_map"][:, :1]) < 2e-5
    r4 = out["reduce4"].cpu()
    assert abs(float(r4[0] / (r4[1] + 1e-5)) - float(out_o["gradient_error"])) < 1e-4


@pytest.mark.parametrize("tag", ["r16c3_", "r64c3_", "r64c1_"])
def test_conv_discriminator_golden_f7(ops, tag):
    g = load_golden("f7_discriminator")
    dsd = sub_sd(g, tag + "w.")
    x = g[tag + "x"].cuda()
    n = len([k for k in dsd if k.startswith("blocks.")])
    for i in range(n):
        x = ops.conv4x4_fwd(x, dsd[f"blocks.{i}.weight"].cuda(), None, 2, 1, 0.2)
    d = ops.conv4x4_fwd(x, dsd["conv_out.weight"].cuda(), None, 1, 0, 1.0).reshape(x.shape[0], -1)
    assert maxdiff(d.cpu(), g[tag + "d"]) < 2e-5


@pytest.mark.parametrize("B,Cin,H,Cout,stride,pad", [(1, 3, 64, 64, 2, 1), (1, 256, 8, 512, 2, 1), (3, 512, 4, 7, 1, 0),
                                                      (2, 5, 10, 33, 2, 1), (4, 1, 128, 32, 2, 1)])
def test_conv4x4_vs_torch(ops, B, Cin, H, Cout, stride, pad):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / math.sqrt(Cin * 16)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad), 0.2)
    y = ops.conv4x4_fwd(x.cuda(), w.cuda(), b.cuda(), stride, pad, 0.2)
    assert maxdiff(y.cpu(), ref) < 2e-5


def test_upfirdn2d_golden_f8(ops):
    g = load_golden("f8_augment")
    f1 = g["Hz_geom"].cuda()
    x = g["ufd_x"].cuda()
    # upsample2d = two separable passes (upfirdn2d.py:239-241, 325-357)
    up = ops.upfirdn2d(x, f1[None, :], upx=2, padx0=6, padx1=5, gain=2.0)
    up = ops.upfirdn2d(up, f1[:, None], upy=2, pady0=6, pady1=5, gain=2.0)
    assert maxdiff(up.cpu(), g["ufd_up"]) < 1e-5
    dn = ops.upfirdn2d(up, f1[None, :], downx=2, padx0=-2 + 5, padx1=-2 + 5, flip=True)
    dn = ops.upfirdn2d(dn, f1[:, None], downy=2, pady0=-2 + 5, pady1=-2 + 5, flip=True)
    assert maxdiff(dn.cpu(), g["ufd_down"]) < 1e-5
    y = ops.upfirdn2d(x, g["ufd_f2d"].cuda(), upx=2, upy=1, downx=1, downy=3, padx0=1, padx1=2, pady0=0, pady1=3, gain=1.7)
    assert maxdiff(y.cpu(), g["ufd_general"]) < 1e-5


def test_affine_grid_sample_and_pad(ops):
    g = torch.Generator().manual_seed(0)
    B, C, Hi, Wi, Ho, Wo = 3, 2, 37, 29, 24, 40
    x = torch.randn(B, C, Hi, Wi, generator=g)
    theta = torch.eye(2, 3).repeat(B, 1, 1) + 0.3 * torch.randn(B, 2, 3, generator=g)
    grid = torch.nn.functional.affine_grid(theta, [B, C, Ho, Wo], align_corners=False)
    ref = torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    y = ops.affine_grid_sample_fwd(x.cuda(), theta.cuda(), Ho, Wo)
    assert maxdiff(y.cpu(), ref) < 2e-5
    gy = torch.randn(B, C, Ho, Wo, generator=g)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.grid_sample(xr, grid, mode="bilinear", padding_mode="zeros", align_corners=False).backward(gy)
    gx = ops.affine_grid_sample_bwd(gy.cuda(), theta.cuda(), Hi, Wi)
    assert maxdiff(gx.cpu(), xr.grad) < 5e-5
    p = (3, 5, 2, 7)
    yp = ops.reflect_pad_fwd(x.cuda(), *p)
    refp = torch.nn.functional.pad(x, list(p), mode="reflect")
    assert maxdiff(yp.cpu(), refp) == 0
    gp = torch.randn_like(refp)
    xr2 = x.clone().requires_grad_(True)
    torch.nn.functional.pad(xr2, list(p), mode="reflect").backward(gp)
    gxp = ops.reflect_pad_bwd(gp.cuda(), Hi, Wi, *p)
    assert maxdiff(gxp.cpu(), xr2.grad) < 1e-5


@pytest.mark.parametrize("wscale,pscale,noise", [(1.0, 1.0, 0.0), (3.0, 1.0, 0.0), (1.0, 3.0, 0.0), (1.0, 1.0, 0.5)])
def test_f16x3_tracks_native_fp32_off_distribution(ops, sdf_sd, col_sd, wscale, pscale, noise):
    """The default operand mode (two scaled fp16 limbs, per-point normalised adjoints) against the native fp32 MFMA
    path on inputs outside the golden vectors' range: 3x latents, points up to |x| = 3 (|d sdf/dx| ~ 35), weights
    with 50 % multiplicative-scale noise.  No overflow, differences at fp32 round-off level."""
    from oi_amd.params import stack_field_params
    g = torch.Generator().manual_seed(7)
    sd = {k: v.clone() for k, v in sdf_sd.items()}
    if noise:
        for k in sd:
            if k.startswith("pts_linears") and k.endswith("weight") and ".gamma." not in k and ".beta." not in k:
                sd[k] = sd[k] + noise * sd[k].abs().mean() * torch.randn(sd[k].shape, generator=g)
    P = stack_field_params(dev(sd), dev(col_sd))
    B, n = 2, 16384
    pts = ((torch.rand(B * n, 3, generator=g) * 2 - 1) * pscale).cuda()
    z = (torch.randn(B, 64, generator=g) * wscale).cuda()
    _, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], z=z)
    res = {}
    for name, prec in (("f32", 0), ("f16x3", 4)):
        pk = ops.mlp_pack_weights(P["w0"], P["b0"], P["wh"], P["bh"], P["wsig"], P["bsig"], P["wv"], P["bv"], P["wrgb"],
                                  P["brgb"], prec)
        res[name] = ops.sdf_mlp_fwd(pts, pk, gamma, beta, B, prec, want_grad=True, want_rgb=True)[:3]
    a, b = res["f32"], res["f16x3"]
    assert all(bool(torch.isfinite(t).all()) for t in b)
    gs = max(1.0, float(a[1].abs().max()))
    assert maxdiff(a[0], b[0]) < 2e-5 and maxdiff(a[1], b[1]) / gs < 2e-5 and maxdiff(a[2], b[2]) < 2e-5


def test_device_sincos_accuracy(ops):
    """The sin / cos of the MLP kernels (whole-period reduction in revolutions + v_sin_f32 / v_cos_f32) against float64
    on the fp32 phase, over the phase range of FiLM-SIREN layers and far beyond it; torch.sin of the reference is
    0.5-ulp accurate, so this error adds directly to the parity budget."""
    import ctypes
    from oi_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(11)
    x = torch.cat([(torch.rand(1 << 20, generator=g) * 2 - 1) * 100.0, (torch.rand(1 << 20, generator=g) * 2 - 1) * 1000.0,
                   torch.linspace(-8.0, 8.0, 1 << 16), torch.tensor([0.0, 1e-8, -1e-8, 3.14159274, -3.14159274, 1.57079637])])
    xd = x.cuda()
    s, c = torch.empty_like(xd), torch.empty_like(xd)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.check(L.oi_selftest_sincos(p(xd), p(s), p(c), x.numel(), 0, ops._stream()), "oi_selftest_sincos")
    es = (s.cpu().double() - torch.sin(x.double())).abs()
    ec = (c.cpu().double() - torch.cos(x.double())).abs()
    assert float(es.max()) < 3e-7 and float(ec.max()) < 3e-7, (float(es.max()), float(ec.max()))
    small = x.abs() <= 100.0
    assert float(es[small].max()) < 2.5e-7 and float(ec[small].max()) < 2.5e-7


def test_cu_slot_exclusive(ops):
    """The register-resident forward kernel parks features in a scratch slot indexed by the physical CU (XCC id + SE/SH/CU
    bits of HW_ID) when a launch has more than 512 workgroups.  On the device: many more workgroups than CUs, each with
    that kernel's LDS footprint, must never find their slot busy, and the ids must spread over many distinct slots."""
    from oi_amd import lib
    L = lib.load()
    busy = torch.zeros(4096, dtype=torch.int32, device="cuda")
    used = torch.zeros(4096, dtype=torch.int32, device="cuda")
    clashes = torch.zeros(1, dtype=torch.int32, device="cuda")
    stream = ops._stream()
    for _ in range(3):
        lib.check(L.oi_selftest_cu_slots(busy.data_ptr(), clashes.data_ptr(), used.data_ptr(), 20000, 200, stream),
                  "oi_selftest_cu_slots")
    torch.cuda.synchronize()
    assert int(clashes) == 0, int(clashes)
    assert int(busy.abs().sum()) == 0
    n_slots = int((used != 0).sum())
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert n_cu // 2 <= n_slots <= n_cu, (n_slots, n_cu)


def test_sdf_mlp_large_launch_uses_cu_slots(ops, packed_all):
    """> 512 workgroups (CU-indexed scratch; here 6001) gives the same result as the same points in small launches."""
    P, packs = packed_all
    g1 = load_golden("f1_film_siren")
    w, gamma, beta = ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], w=g1["w"][:1].cuda())
    n = 128 * 6000 + 37
    gen = torch.Generator(device="cuda").manual_seed(3)
    pts = (torch.rand(n, 3, device="cuda", generator=gen) * 2 - 1) * 1.1
    sdf, grad, rgb, _, _ = ops.sdf_mlp_fwd(pts, packs["f16x3"], gamma, beta, 1, 4, want_grad=True, want_rgb=True)
    for lo in (0, 128 * 3000 + 5, n - 4096):
        sl = slice(lo, lo + 4096)
        s2, g2, r2, _, _ = ops.sdf_mlp_fwd(pts[sl].contiguous(), packs["f16x3"], gamma, beta, 1, 4, want_grad=True, want_rgb=True)
        # different tile alignment -> different point-to-lane mapping only; per-point arithmetic is identical
        assert maxdiff(s2, sdf[sl]) == 0 and maxdiff(g2, grad[sl]) == 0 and maxdiff(r2, rgb[sl]) == 0


@pytest.mark.parametrize("act,grad", [(1, 0), (1, 1), (1, 2), (3, 0), (3, 1), (3, 2)])
def test_fused_bias_act_plugin_signature(act, grad):
    """VERDICT r1 missing #8: the stand-alone op under the reference plugin's own argument list
    (stylesdf/op/fused_bias_act.cpp:11-20; semantics restated from fused_bias_act_kernel.cu:18-49)."""
    from oi_amd.plugin_ops import fused_bias_act
    g = torch.Generator().manual_seed(10 * act + grad)
    x = torch.randn(3, 5, 7, 6, generator=g).cuda()
    b = torch.randn(5, generator=g).cuda()
    ref = torch.randn(3, 5, 7, 6, generator=g).cuda()
    alpha, scale = 0.2, 2 ** 0.5
    for bias, refer in ((b, ref), (x.new_empty(0), ref), (b, x.new_empty(0))):
        out = fused_bias_act(x, bias, refer, act, grad, alpha, scale)
        v = x + (bias.view(1, -1, 1, 1) if bias.numel() else 0.0)
        r = refer if refer.numel() else torch.zeros_like(x)
        if grad == 2:
            want = torch.zeros_like(x)
        elif act == 1:
            want = v
        elif grad == 0:
            want = torch.where(v > 0, v, v * alpha)
        else:
            want = torch.where(r > 0, v, v * alpha)
        assert torch.equal(out, want * scale)
    x2 = torch.randn(4, 9, generator=g).cuda()      # 2-D input (the mapping network's use): bias over dim 1, step 1
    b2 = torch.randn(9, generator=g).cuda()
    v = x2 + b2
    assert torch.equal(fused_bias_act(x2, b2, x2.new_empty(0), 3, 0, 0.2, 1.0), torch.where(v > 0, v, v * 0.2))


def test_grid_sample_plugin_vs_aten_fwd_bwd_double_bwd():
    """grid_sample(input, grid) of grid_sample_gradfix.py:33-66 (bilinear / zeros / align_corners=False) against ATen's
    own kernels on the same device: value, both first derivatives, and the double backward through grad_input that the
    R1 penalty needs."""
    import torch.nn.functional as F
    from oi_amd.plugin_ops import grid_sample
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 3, 9, 11, generator=g).cuda()
    grid = (torch.rand(2, 6, 5, 2, generator=g) * 2.6 - 1.3).cuda()      # a third of the samples fall outside
    cot = torch.randn(2, 3, 6, 5, generator=g).cuda()
    xa, ga = x.clone().requires_grad_(True), grid.clone().requires_grad_(True)
    xb, gb = x.clone().requires_grad_(True), grid.clone().requires_grad_(True)
    ya = grid_sample(xa, ga)
    yb = F.grid_sample(xb, gb, mode="bilinear", padding_mode="zeros", align_corners=False)
    assert maxdiff(ya, yb) < 2e-6
    gxa, gga = torch.autograd.grad((ya * cot).sum(), [xa, ga])
    gxb, ggb = torch.autograd.grad((yb * cot).sum(), [xb, gb])
    assert maxdiff(gxa, gxb) < 2e-6
    assert maxdiff(gga, ggb) < 2e-5 * max(1.0, float(ggb.abs().max()))
    # double backward (ATen has none for grid_sampler_2d_backward -- the reason the reference wraps it): grad_input is
    # linear in the cotangent, so d <grad_input(c), v> / dc is the forward op applied to v
    v = torch.randn_like(x)
    ca = cot.clone().requires_grad_(True)
    ga1, = torch.autograd.grad((grid_sample(xa, ga.detach()) * ca).sum(), xa, create_graph=True)
    da, = torch.autograd.grad((ga1 * v).sum(), ca)
    db = F.grid_sample(v, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    assert maxdiff(da, db) < 2e-6


@pytest.mark.parametrize("B,Cin,H,Cout,x_slope,tile", [(16, 64, 32, 128, 1.0, 64), (9, 128, 32, 128, 0.2, 64), (64, 64, 32, 128, 0.2, 128),
                                                       (37, 16, 20, 192, 1.0, 64)])
def test_conv4x4_tiled_f16x3_vs_fp64(ops, B, Cin, H, Cout, x_slope, tile):
    """The LDS-tiled F16X3 convolution (M = B * Ho * Wo >= 2048 pixels; both pixel-tile widths, ragged last tile, odd
    image size, LeakyReLU-on-load) against an fp64 convolution: operands carry 22 mantissa bits, accumulation is fp32 with a
    fixed summation order -- bit-identical between runs."""
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / math.sqrt(Cin * 16)
    b = torch.randn(Cout, generator=g)
    Ho = (H + 2 - 4) // 2 + 1
    assert B * Ho * Ho >= 2048 and (math.ceil(B * Ho * Ho / 128) * (Cout // 64) < 256) == (tile == 64)
    xin = torch.nn.functional.leaky_relu(x.double(), x_slope) if x_slope != 1.0 else x.double()
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xin, w.double(), b.double(), stride=2, padding=1), 0.2)
    y = ops.conv4x4_fwd(x.cuda(), w.cuda(), b.cuda(), 2, 1, 0.2, x_slope=x_slope)
    assert maxdiff(y.cpu(), ref) < 5e-6 * max(1.0, float(ref.abs().max()))
    y2 = ops.conv4x4_fwd(x.cuda(), w.cuda(), b.cuda(), 2, 1, 0.2, x_slope=x_slope)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("up,down,taps,pad", [((3, 1), (2, 1), (7, 1), (4, 3, 0, 0)), ((1, 3), (1, 2), (1, 5), (0, 0, 2, 5)),
                                               ((2, 3), (3, 2), (5, 4), (3, 1, 2, 4)), ((1, 1), (1, 1), (13, 1), (6, 6, 0, 0)),
                                               ((2, 1), (1, 1), (12, 1), (-2, 7, 0, 0))])
def test_upfirdn2d_general_factors_vs_dense_restatement(ops, up, down, taps, pad):
    """upfirdn2d beyond the ADA shapes (up / down factors of 3, 2-D filters, a 13-tap filter that leaves the 12-tap fast
    path, negative padding = cropping): zero-insert upsample -> pad / crop -> correlate with the flipped filter -> decimate,
    restated densely with torch ops (upfirdn2d.py:160-200 of the reference's ADA copy)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(up) * 7 + sum(taps))
    B, C, H, W = 2, 3, 11, 13
    x = torch.randn(B, C, H, W, generator=g)
    f = torch.randn(taps[1], taps[0], generator=g)             # [fh, fw]
    upx, upy = up
    dx, dy = down
    px0, px1, py0, py1 = pad
    u = torch.zeros(B, C, H * upy, W * upx, dtype=torch.float64)
    u[:, :, ::upy, ::upx] = x.double()
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0):u.shape[2] - max(-py1, 0), max(-px0, 0):u.shape[3] - max(-px1, 0)]
    w = f.double().flip([0, 1])[None, None].repeat(C, 1, 1, 1)
    ref = F.conv2d(u, w, groups=C)[:, :, ::dy, ::dx]
    y = ops.upfirdn2d(x.cuda(), f.cuda(), upx, upy, dx, dy, px0, px1, py0, py1)
    assert tuple(y.shape) == tuple(ref.shape), (y.shape, ref.shape)
    assert maxdiff(y.cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_conv_double_backward_operands_of_gradient_scale_stay_exact(ops):
    """The R1 double backward runs gradients (1e-5-scale here) through the forward convolution entry (_Dgrad.backward ->
    _Conv.apply(ggx, w)) at a batch that would select the large-batch kernel with its UNSCALED fp16 limbs; the autograd node
    asks for OI_CONV_ANY_SCALE and must match an fp64 convolution to fp32 accuracy at any operand scale (advisor, round 2)."""
    from oi_amd.autograd_conv import _Conv
    g = torch.Generator().manual_seed(3)
    B, Cin, H, Cout = 64, 64, 16, 128
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / math.sqrt(Cin * 16)
    for scale in (1.0, 1e-5, 1e-8):
        x = torch.randn(B, Cin, H, H, generator=g) * scale
        ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=1)
        y = _Conv.apply(x.cuda(), w.cuda(), 2, 1)
        assert maxdiff(y.cpu(), ref) < 5e-6 * float(ref.abs().max()), (scale, maxdiff(y.cpu(), ref), float(ref.abs().max()))
    # the same call WITHOUT the flag takes the tiled kernel and is only good for O(1) operands -- which is all it is used for
    y = ops.conv4x4_fwd((x / scale).cuda(), w.cuda(), None, 2, 1, 1.0)
    ref = torch.nn.functional.conv2d((x / scale).double(), w.double(), stride=2, padding=1)
    assert maxdiff(y.cpu(), ref) < 5e-6 * float(ref.abs().max())


def test_upfirdn2d_more_planes_than_one_launch_grid(ops):
    """B * C = 65,539 planes (a 128 x 512-channel StyleGAN2 layer is 65,536): grid.z holds 65,535, the rest goes in a second
    launch (advisor, round 2)."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 65539, 4, 4, generator=g)
    f = torch.tensor([1.0, 3.0, 3.0, 1.0]) / 8
    y = ops.upfirdn2d(x.cuda(), f[None].cuda(), 2, 1, 1, 1, 2, 1, 0, 0)
    import torch.nn.functional as F
    u = torch.zeros(1, 65539, 4, 8, dtype=torch.float64)
    u[..., ::2] = x.double()
    ref = F.conv2d(F.pad(u, [2, 1, 0, 0]).view(65539, 1, 4, 11), f.double().flip(0).view(1, 1, 1, 4)).view(1, 65539, 4, 8)
    assert tuple(y.shape) == (1, 65539, 4, 8)
    assert maxdiff(y.cpu(), ref) < 1e-6


@pytest.mark.parametrize("B,C,R,static", [(2, 3, 64, True), (3, 1, 32, False), (1, 3, 128, True)])
def test_ada_geom_fused_matches_the_staged_chain(ops, B, C, R, static, monkeypatch):
    """oi_ada_geom_fwd (pad + up-FIR | resample + down-FIR: two launches) against the four separate stages it replaces, for
    translations, scales and a rotation, with fitted and with static margins: values, the gradient to the images (the fused
    node's backward is the adjoint chain) and the double backward an R1 penalty takes."""
    import oi_amd.augment as A
    aug = A.AugmentPipe(xint=1, scale=1, rotate=1).cuda()
    np.random.seed(B * 7 + R)
    x = torch.rand(B, C, R, R, generator=torch.Generator().manual_seed(R)).cuda()
    G = aug.sample_G_inv(x)
    margins = aug.static_margins(R, R) if static else aug.margins_for(G, R, R)
    theta = torch.from_numpy(aug.theta_for(G, margins, R, R)).cuda()
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(A, "FUSED", fused)
        xi = x.clone().requires_grad_()
        y = aug.apply_theta(xi, theta, margins)
        cot = torch.rand(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
        (gx,) = torch.autograd.grad((y * cot).sum(), xi, create_graph=True)
        # R1-style second order: d/dx of |gx|^2 contracted with another direction goes back through the forward ops
        xi2 = x.clone().requires_grad_()
        y2 = aug.apply_theta(xi2, theta, margins)
        (g1,) = torch.autograd.grad(y2.square().sum(), xi2, create_graph=True)
        (g2,) = torch.autograd.grad(g1.square().sum(), xi2)
        outs.append((y.detach(), gx.detach(), g2.detach()))
    for name, a, b in zip(("value", "gradient", "double backward"), outs[0], outs[1]):
        assert maxdiff(a, b) < 2e-6 * max(1.0, float(b.abs().max())), (name, maxdiff(a, b), float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,static,R", [(2, 3, True, 64), (5, 3, False, 64), (3, 1, False, 64), (64, 3, False, 64), (2, 3, True, 128),
                                          (3, 1, False, 128)])
def test_ada_geom_separable_matches_the_two_launch_form(ops, B, C, static, R, monkeypatch):
    """oi_ada_geom_sep_fwd (one launch, y = A_y x A_x^T: what a pipe without rotations takes at 64 x 64 and 128 x 128) against oi_ada_geom_fwd
    for flips, integer and fractional translations, isotropic and anisotropic scales, fitted and static margins: values, the
    gradient to the images and the R1-style double backward (whose second pass is this forward again)."""
    import oi_amd.augment as A
    import oi_amd.ops as O
    aug = A.AugmentPipe(xflip=1, xint=1, scale=1, aniso=1, xfrac=1).cuda()
    np.random.seed(B * 11 + C)
    x = torch.rand(B, C, R, R, generator=torch.Generator().manual_seed(B)).cuda()
    G = aug.sample_G_inv(x)
    assert np.all(G[:, 0, 1] == 0) and np.all(G[:, 1, 0] == 0) and np.any(G[:, 0, 0] < 0) or B < 5   # (a flip among the draws)
    margins = aug.static_margins(R, R) if static else aug.margins_for(G, R, R)
    theta = torch.from_numpy(aug.theta_for(G, margins, R, R)).cuda()
    assert float(theta[:, 0, 1].abs().max()) == 0.0 and float(theta[:, 1, 0].abs().max()) == 0.0
    outs = []
    for sep in (True, False):
        monkeypatch.setattr(O, "ADA_SEPARABLE", sep)
        xi = x.clone().requires_grad_()
        y = aug.apply_theta(xi, theta, margins)
        cot = torch.rand(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
        (gx,) = torch.autograd.grad((y * cot).sum(), xi, create_graph=True)
        xi2 = x.clone().requires_grad_()
        y2 = aug.apply_theta(xi2, theta, margins)
        (g1,) = torch.autograd.grad(y2.square().sum(), xi2, create_graph=True)
        (g2,) = torch.autograd.grad(g1.square().sum(), xi2)
        outs.append((y.detach(), gx.detach(), g2.detach()))
    if B <= 5:   # both against the staged chain in float64 (the oracle's own stages; its sampling coordinates are float64 too)
        f64 = aug.Hz_geom.double().cpu()
        mx0, my0, mx1, my1 = margins
        xp = torch.nn.functional.pad(x.double().cpu(), [mx0, mx1, my0, my1], mode="reflect")
        g = O_ref.affine_bilinear_sample(O_ref.upsample2d(xp, f64), theta.double().cpu(), 2 * (R + 6), 2 * (R + 6))
        ref = O_ref.downsample2d(g, f64, down=2, padding=-6, flip=True)
        for tag, got in (("separable", outs[0][0]), ("two-launch", outs[1][0])):
            record_margin("ada_geom_vs_float64_stages", f"{tag} [{B}-{C}-{int(static)}-{R}]", maxdiff(got.cpu(), ref))
            # (fp32 sampling coordinates on a canvas of up to 2 (3 R - 2) pixels: 1.3e-5 measured at 64, 2.7e-5 at 128, both forms)
            assert maxdiff(got.cpu(), ref) < 2e-5 * R / 64, (tag, maxdiff(got.cpu(), ref))
    for name, a, b in zip(("value", "gradient", "double backward"), outs[0], outs[1]):
        record_margin("ada_geom_separable_vs_two_launch", f"{name} [{B}-{C}-{int(static)}-{R}]", maxdiff(a, b) / max(1.0, float(b.abs().max())))
        assert maxdiff(a, b) < 1.5e-6 * max(1.0, float(b.abs().max())), (name, maxdiff(a, b), float(b.abs().max()))   # (measured: <= 4.9e-7)
    assert maxdiff(outs[0][0], outs[1][0]) > 0.0 or B == 0   # (two different kernels ran: not bit-identical by construction)
    # a rotation in the pipe: the general form, whatever the switch says
    monkeypatch.setattr(O, "ADA_SEPARABLE", True)
    rot = A.AugmentPipe(xint=1, rotate=1).cuda()
    Gr = rot.sample_G_inv(x)
    mr = rot.margins_for(Gr, R, R)
    tr = torch.from_numpy(rot.theta_for(Gr, mr, R, R)).cuda()
    ya = rot.apply_theta(x, tr, mr)
    monkeypatch.setattr(O, "ADA_SEPARABLE", False)
    assert torch.equal(ya, rot.apply_theta(x, tr, mr))


@pytest.mark.gpu
@pytest.mark.parametrize("R,C,B", [(64, 1, 70), (128, 3, 3)])
def test_ada_geom_separable_host_matrices_equal_device_matrices(ops, R, C, B):
    """oi_ada_geom_sep_fwd with the sampling matrices by value (64 images per launch: batch 70 = two launches) against the same
    matrices read from device memory: the same kernel arithmetic, bit-identical; the adjoint entry satisfies <A x, g> = <x, A^T g>."""
    import oi_amd.augment as A
    import oi_amd.ops as OPS
    aug = A.AugmentPipe(xint=1, scale=1).cuda()
    np.random.seed(R + B)
    x = torch.rand(B, C, R, R, generator=torch.Generator().manual_seed(2)).cuda()
    m = aug.static_margins(R, R)
    th = aug.theta_fast(B, R, R)
    y_host = OPS.ada_geom_sep_host(x, th, aug.Hz_geom, m)
    thd = torch.from_numpy(th).cuda()
    y_dev = OPS.ada_geom_fwd(x, thd, aug.Hz_geom, m, axis_aligned=True)
    assert torch.equal(y_host, y_dev)
    g = torch.rand(y_dev.shape, generator=torch.Generator().manual_seed(3)).cuda()
    gx = OPS.ada_geom_adj_sep(g, thd, aug.Hz_geom, m)
    lhs, rhs = float((y_dev.double() * g.double()).sum()), float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) < 2e-6 * abs(lhs), (lhs, rhs)
    assert torch.equal(gx, OPS.ada_geom_adj_sep(g, thd, aug.Hz_geom, m))   # (fixed-point build: bit-reproducible)
    bad = th.copy()
    bad[0, 0, 1] = 1e-3
    with pytest.raises(ValueError):
        OPS.ada_geom_sep_host(x, bad, aug.Hz_geom, m)


def test_gan_losses_fused_match_the_reference_composition():
    """oi_gan_losses_fwd / _bwd (one launch each way) against GANLoss + compute_grad2 + PositionLoss summed as the trainer
    sums them (src/loss/gan.py:5-22, 39-49; src/loss/position.py:4-18; gan_pose_trainer.py:163-190), through a toy
    discriminator whose R1 double backward reaches its weights."""
    from oi_amd.losses import GANLoss, PositionLoss, compute_grad2, gan_losses, grad_wrt_input
    torch.manual_seed(3)
    B, K, n = 3, 7, 3 * 8 * 8
    w = (torch.randn(K, n, device="cuda") * 0.2).requires_grad_()
    net = lambda x: torch.tanh(x.reshape(B, -1) @ w.t()) * 3.0
    xr = torch.randn(B, 3, 8, 8, device="cuda")
    xf = torch.randn(B, 3, 8, 8, device="cuda")
    pose = torch.randn(B, K - 1, device="cuda")
    aux_w, reg_w = 0.37, 10.0
    gan, mse = GANLoss("bce"), PositionLoss("mse")

    x1 = xr.clone().requires_grad_()
    d_real, d_fake = net(x1), net(xf)
    real, fake = gan(d_real[:, :1], 1), gan(d_fake[:, :1], 0)
    reg, aux = compute_grad2(d_real[:, :1], x1), mse(d_fake[:, 1:], pose)
    ref = real + fake + reg_w * reg + aux_w * aux
    (gw_ref,) = torch.autograd.grad(ref, w)

    x2 = xr.clone().requires_grad_()
    d_real, d_fake = net(x2), net(xf)
    gx = grad_wrt_input(d_real[:, :1], x2)
    total, parts = gan_losses(d_real, d_fake, pose, gx, torch.full((), aux_w, device="cuda"), reg_w)
    (gw,) = torch.autograd.grad(total, w)
    for got, want in ((total, ref), (parts[0], real + fake), (parts[1], reg), (parts[2], fake), (parts[3], real), (parts[4], aux)):
        assert abs(float(got.detach()) - float(want.detach())) < 2e-6 * max(1.0, abs(float(want.detach()))), (float(got.detach()), float(want.detach()))
    assert float((gw - gw_ref).abs().max()) < 2e-6 * float(gw_ref.abs().max())
    # generator-step use: BCE against 1 alone, gradient w.r.t. the logits only in column 0
    d = net(xf).detach().requires_grad_()
    t2, _ = gan_losses(d_real=d)
    (gd,) = torch.autograd.grad(t2, d)
    d3 = d.detach().clone().requires_grad_()
    (gd_ref,) = torch.autograd.grad(gan(d3[:, :1], 1), d3)
    assert abs(float(t2.detach()) - float(gan(d3[:, :1], 1).detach())) < 1e-6 and float((gd - gd_ref).abs().max()) < 1e-7


@pytest.mark.parametrize("B,Cin,Cout,H,stride,pad", [(1, 32, 64, 32, 2, 1), (3, 5, 7, 9, 2, 1), (2, 256, 7, 4, 1, 0)])
def test_conv_gradients_masked_on_load_and_accumulated_match_the_staged_ops(B, Cin, Cout, H, stride, pad):
    """oi_conv4x4_dgrad_masked / oi_conv4x4_wgrad_masked: LeakyReLU mask of the forward output applied to the incoming
    gradient on load == oi_lrelu_mask_mul followed by the plain kernels; accumulate adds to what the buffer holds."""
    from oi_amd import ops
    torch.manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.1
    Ho = (H + 2 * pad - 4) // stride + 1
    g = torch.randn(B, Cout, Ho, Ho, device="cuda")
    y = torch.randn(B, Cout, Ho, Ho, device="cuda")
    gm = ops.lrelu_mask_mul(g, y, 0.2)
    for got, want in ((ops.conv4x4_dgrad(g, w, H, H, stride, pad, y, 0.2), ops.conv4x4_dgrad(gm, w, H, H, stride, pad)),
                      (ops.conv4x4_wgrad(g, x, stride, pad, y, 0.2), ops.conv4x4_wgrad(gm, x, stride, pad))):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())   # (split-K atomics: order of summation)
    base = torch.randn(Cout, Cin, 4, 4, device="cuda")
    acc = base.clone()
    assert ops.conv4x4_wgrad(g, x, stride, pad, y, 0.2, acc=acc) is acc
    want = base + ops.conv4x4_wgrad(gm, x, stride, pad)
    assert float((acc - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("B,Cin,Cout,H,stride,pad", [(1, 32, 64, 32, 2, 1), (3, 5, 7, 9, 2, 1), (2, 256, 7, 4, 1, 0)])
def test_conv_backward_in_one_launch_matches_the_two_kernels(B, Cin, Cout, H, stride, pad):
    """oi_conv4x4_bwd_masked (data + weight gradient, one launch) == oi_conv4x4_dgrad_masked + oi_conv4x4_wgrad_masked."""
    from oi_amd import ops
    torch.manual_seed(B + Cout)
    x = torch.randn(B, Cin, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.1
    Ho = (H + 2 * pad - 4) // stride + 1
    g, y = torch.randn(B, Cout, Ho, Ho, device="cuda"), torch.randn(B, Cout, Ho, Ho, device="cuda")
    for ref in (None, y):
        gx, gw = ops.conv4x4_bwd(g, w, x, stride, pad, ref, 0.2)
        gx0, gw0 = ops.conv4x4_dgrad(g, w, H, H, stride, pad, ref, 0.2), ops.conv4x4_wgrad(g, x, stride, pad, ref, 0.2)
        assert float((gx - gx0).abs().max()) <= 1e-5 * float(gx0.abs().max())
        assert float((gw - gw0).abs().max()) <= 1e-5 * float(gw0.abs().max())
    base = torch.randn_like(w)
    acc = base.clone()
    _, out = ops.conv4x4_bwd(g, w, x, stride, pad, y, 0.2, acc=acc)
    assert out is acc and float((acc - (base + gw0)).abs().max()) <= 1e-5 * float((base + gw0).abs().max())


def test_weighted_sum_and_render_scalars_match_tensor_ops():
    """Round 3 glue kernels: the weighted sum of scalar loss terms and the renderer's two derived scalars, forward and
    gradient, against the tensor-op composition they replace (gan_pose_trainer.py:122-137, renderer.py:430-446)."""
    from oi_amd.losses import weighted_sum
    from oi_amd.renderer import render_scalars
    g = torch.Generator().manual_seed(3)
    vals = [torch.randn((), generator=g).cuda().requires_grad_() for _ in range(5)]
    vals[3] = torch.randn(1, generator=g).cuda().requires_grad_()   # a 1-element term keeps its shape in the gradient
    ws = [1.0, 0.1, 10.0, 0.0, -2.5]
    total = weighted_sum(vals, ws)
    ref = sum(v.detach().double().reshape(()) * w for v, w in zip(vals, ws))
    assert abs(float(total) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    (total * 3.0).backward()
    for v, w in zip(vals, ws):
        assert v.grad.shape == v.shape and abs(float(v.grad.reshape(())) - 3.0 * w) < 1e-6
    r4 = torch.tensor([12.5, 340.0, 7.25, 0.0], device="cuda", requires_grad=True)
    r4b = r4.detach().clone().requires_grad_()
    err, surf = render_scalars(r4, 4096)
    err_o, surf_o = r4b[0] / (r4b[1] + 1e-5), r4b[2] / 4096.0
    assert maxdiff(err, err_o) < 1e-7 and maxdiff(surf, surf_o) < 1e-9
    (2.0 * err + 5.0 * surf).backward()
    (2.0 * err_o + 5.0 * surf_o).backward()
    assert maxdiff(r4.grad, r4b.grad) < 1e-8
    r4.grad = None
    render_scalars(r4, 4096)[0].backward()          # only one of the two outputs used: the other gradient is absent
    assert abs(float(r4.grad[0]) - 1.0 / (340.0 + 1e-5)) < 1e-9 and float(r4.grad[2]) == 0.0


def test_stage_inputs_copies_and_immediates(ops):
    """oi_stage_inputs: several device copies and up to 64 host floats in one launch (inputs of a captured step)."""
    g = torch.Generator().manual_seed(4)
    srcs = [torch.randn(n, generator=g).cuda() for n in (1, 4099, 3 * 64 * 64)]
    dsts = [torch.full_like(s, -7.0) for s in srcs]
    imm = np.arange(13, dtype=np.float32) * 0.37 - 1.0
    flat = torch.full((16,), -7.0, device="cuda")
    ops.stage_inputs(list(zip(srcs, dsts)) + [(None, None)], imm, flat)
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    assert torch.equal(flat[:13].cpu(), torch.from_numpy(imm)) and bool((flat[13:] == -7.0).all())
    ops.stage_inputs([(srcs[1] * 2, dsts[1])])      # copies only
    assert torch.equal(dsts[1], srcs[1] * 2)


def test_prezeroed_declaration_is_per_stream(ops):
    """the boundary is re-entrant and thread-safe.  While one thread works inside a ZeroPool (the library
    skips the fills of accumulate-outputs launched on THAT stream), a second thread on its own stream calls
    oi_conv4x4_wgrad into memory full of garbage and must get the right sums -- the round-3 flag was process-wide and the
    second thread would have accumulated onto the garbage."""
    import threading
    from oi_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 16, generator=g).cuda()
    gy = torch.randn(2, 32, 8, 8, generator=g).cuda()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (32, 16, 4, 4), gy.double().cpu(), stride=2, padding=1)
    torch.cuda.synchronize()
    s_main, s_other = torch.cuda.Stream(), torch.cuda.Stream()
    inside, release, result = threading.Event(), threading.Event(), {}

    def pool_owner():
        with torch.cuda.stream(s_main):
            pool = ops.ZeroPool()
            for _ in range(2):   # the first pass measures, the second hands out pool memory
                pool.begin(x.device)
                with pool:
                    gw = ops.conv4x4_wgrad(gy, x, 2, 1)
                    if pool.buf is not None:
                        assert gw.data_ptr() >= pool.buf.data_ptr() and gw.data_ptr() < pool.buf.data_ptr() + pool.buf.numel() * 4
                        inside.set()
                        assert release.wait(60)
            s_main.synchronize()
            result["owner"] = gw.detach().cpu()

    def bystander():
        assert inside.wait(60)
        try:
            with torch.cuda.stream(s_other):
                junk = torch.full((32, 16, 4, 4), 1.0e9, device="cuda")   # the allocator hands this block out again below
                s_other.synchronize()
                ptr = junk.data_ptr()
                del junk
                gw = ops.conv4x4_wgrad(gy, x, 2, 1)
                s_other.synchronize()
                result["reused_garbage_block"] = gw.data_ptr() == ptr
                result["bystander"] = gw.detach().cpu()
                # and through the raw C-ABI, into memory this thread filled itself
                raw = torch.full((32, 16, 4, 4), -7.0e8, device="cuda")
                s_other.synchronize()
                lib.check(L.oi_conv4x4_wgrad(ops._p(gy), ops._p(x), ops._p(raw), 2, 16, 16, 16, 32, 2, 1, ops._stream()), "oi_conv4x4_wgrad")
                s_other.synchronize()
                result["raw"] = raw.cpu()
        finally:
            release.set()

    ta, tb = threading.Thread(target=pool_owner), threading.Thread(target=bystander)
    ta.start(); tb.start(); ta.join(120); tb.join(120)
    assert "owner" in result and "bystander" in result and "raw" in result, result.keys()
    for k in ("owner", "bystander", "raw"):
        err = float((result[k].double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (k, err)
    # the declaration is gone with the pool
    assert L.oi_outputs_prezeroed_stream(ops._vp(s_main.cuda_stream), 0) == 0


# ---------------------------------------------------------------- round 4: fused small launches are bit-identical to the chains
def _ray_batch(N, seed):
    g = torch.Generator().manual_seed(seed)
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.15 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    return ro.cuda().contiguous(), rd.cuda().contiguous(), near.cuda(), far.cuda(), g


@pytest.mark.parametrize("N,Sc,n_new", [(1, 16, 16), (37, 64, 64), (130, 40, 96), (64, 112, 16)])
def test_upsample_mid_matches_upsample_then_midpoints(ops, N, Sc, n_new):
    """oi_upsample_mid (the render's last up-sampling step + the section mid-points in one launch) against oi_upsample
    followed by oi_midpoints: every output bit for bit, ragged ray counts, n_new > Sc included."""
    ro, rd, near, far, g = _ray_batch(N, N + Sc)
    z, _ = ops.coarse_samples(ro, rd, near, far, Sc, torch.rand(N, 1, generator=g).cuda())
    sdf = (torch.rand(N, Sc, generator=g) - 0.4).cumsum(-1).neg().add(2.0).cuda() * 0.1
    zn_a, pn_a, zm_a = ops.upsample(ro, rd, z, sdf, n_new, 64.0, merge=True)
    d_a, m_a, p_a = ops.midpoints(ro, rd, zm_a, 2.0 / Sc)
    zn_b, pn_b, zm_b, (d_b, m_b, p_b) = ops.upsample(ro, rd, z, sdf, n_new, 64.0, mid_last_dist=2.0 / Sc)
    for a, b in ((zn_a, zn_b), (pn_a, pn_b), (zm_a, zm_b), (d_a, d_b), (m_a, m_b), (p_a, p_b)):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("N,T", [(3, 16), (4096, 128), (1001, 70)])
def test_composite_last_block_statistics_match_render_stats(ops, N, T, monkeypatch):
    """The compositing launch's last workgroup sums the per-block partials itself (agent-scope hand-off inside ONE launch):
    reduce4 / ray sums / finals bit-identical to the separate oi_render_stats launch, repeatedly (the arrival counter
    resets itself), and the planar `image` equals the (N, 3) form."""
    g = torch.Generator().manual_seed(N)
    B = 1
    ro, rd, near, far, _ = _ray_batch(N, N)
    z = torch.sort(near.cpu() + (far - near).cpu() * torch.rand(N, T, generator=g), -1).values.cuda()
    dists, mid_z, _ = ops.midpoints(ro, rd, z, 2.0 / T)
    sdf = (1.0 - mid_z + 0.05 * torch.randn(N, T, generator=g).cuda()) * 0.3
    grad = torch.nn.functional.normalize(torch.randn(N, T, 3, generator=g), dim=-1).cuda() * 1.1
    rgb = torch.rand(N, T, 3, generator=g).cuda()
    light = torch.tensor([-0.7, 0.2, 8.0]).cuda()
    args = (sdf, grad, rgb, dists, mid_z, ro, rd, torch.tensor([[0.2, -0.4, -0.9]]).cuda(), torch.rand(B, 3, generator=g).cuda(),
            torch.tensor(0.3).cuda(), light, 0.5, B)
    monkeypatch.setattr(ops, "FUSED_STATS", False)
    ref = ops.composite_fwd(*args)
    monkeypatch.setattr(ops, "FUSED_STATS", True)
    for rep in range(3):
        out = ops.composite_fwd(*args, image_planar=(rep == 1))
        torch.cuda.synchronize()
        for k in ("reduce4", "ray_sums", "finals", "weights", "color_fine", "mask"):
            assert torch.equal(out[k], ref[k]), (k, rep)
        img = out["image"] if rep != 1 else out["image"].view(B, 3, N // B).permute(0, 2, 1).reshape(N, 3)
        assert torch.equal(img, ref["image"]), rep
    assert int(ops._stats_ticket(sdf.device)[0]) == 0


def test_scalar_glue_and_small_upload():
    """oi_scalar_glue: the 0-dim glue of a forward in one launch (renderer.py:404,448; lighting.py:50-60) against the tensor
    expressions it replaces; ops.upload_small: values reach the device through the arguments of one launch."""
    from oi_amd import ops
    for var, amb, spec, shin in ((0.3, -0.66, 0.01, 10.0), (-2.0, 1.5, -0.2, 3.0), (1.5, 0.0, 0.7, 30.0)):
        t = [torch.tensor(v, device="cuda") for v in (var, amb, spec, shin)]
        out5, packed = ops.scalar_glue(*t)
        inv_s = torch.exp(t[0] * 10.0).clamp(1e-6, 1e6)
        ref = torch.stack([inv_s, 1.0 / inv_s, torch.sigmoid(t[1]), 1 - torch.sigmoid(t[1]), t[2].clamp(min=0)])
        assert torch.allclose(out5, ref, rtol=2e-6, atol=0), (out5, ref)
        assert torch.equal(packed, torch.stack(t[1:]))
    vals = np.random.default_rng(0).normal(size=(3, 2, 3)).astype(np.float32)
    up = ops.upload_small(vals, torch.device("cuda"))
    assert up.shape == (3, 2, 3) and np.array_equal(up.cpu().numpy(), vals)
