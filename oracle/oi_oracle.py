"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of the object-intrinsics hot path.

This file is the checker for the HIP kernels in `object-intrinsics_amd/csrc`.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product
package (`oi_amd`) never does, and fails loudly when the HIP library is missing.

It restates, as plain device/dtype-agnostic torch functions (so fp64 runs and autograd-based
checks of first and second order gradients are possible), the algorithm of the reference files
cited per function (paths relative to /root/reference).  It is *pinned* by
`tests/test_oracle_golden.py` against the golden vectors in `tests/golden/*.npz`, which were
produced by importing the reference itself on CPU (`oracle/gen_golden.py`, fixtures F1-F9 of
SURVEY.md section 8c).  The reference's own test-suite holds no vectors for this path.

Conventions: `sd` arguments are dicts with the reference's `state_dict()` key names
(SURVEY.md section 8b); "rows" of point tensors are ordered batch-element-major, i.e. row r of
an (n, .) tensor belongs to batch element r // (n / B)   (src/models/fields.py:55).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# a1/a2: style MLP and FiLM parameters
# --------------------------------------------------------------------------------------


def style_mlp(sd, z):
    """w = 3 x lrelu_0.2(z W^T + b) * 1.
    src/models/fields.py:15-21, src/third_party/stylesdf/model.py:49-54,
    stylesdf/op/fused_act.py:104-116 (scale=1)."""
    h = z
    for i in range(3):
        h = F.leaky_relu(h @ sd[f"style.{i}.weight"].t() + sd[f"style.{i}.bias"], 0.2)
    return h


def film_params(sd, prefix, w):
    """gamma = 15*(w Wg^T + bg) + 30, beta = 0.25*(w Wb^T + bb); (B, C) each.
    src/third_party/stylesdf/volume_renderer.py:27-30, 47-48, 56-57."""
    g = 15.0 * (w @ sd[prefix + "gamma.weight"].t() + sd[prefix + "gamma.bias"]) + 30.0
    b = 0.25 * (w @ sd[prefix + "beta.weight"].t() + sd[prefix + "beta.bias"]) + 0.0
    return g, b


def _per_point(v, n):
    """(B, C) per-element vector -> (n, C) rows, element-major."""
    B = v.shape[0]
    assert n % B == 0, (n, B)
    return v.repeat_interleave(n // B, dim=0)


def n_sdf_layers(sd):
    return len([k for k in sd if k.startswith("pts_linears.") and k.endswith(".weight")
                and k.count(".") == 2])


# --------------------------------------------------------------------------------------
# a3/a4/a5: FiLM-SIREN SDF network, analytic gradient
# --------------------------------------------------------------------------------------


def sdf_forward(sd, x, w, want_grad=False):
    """8 x sin(gamma*(a W^T + b) + beta), then sigma_linear.  Returns sdf (n,1), feat (n,C)
    [, grad (n,3) = d sdf / d x by the analytic reverse sweep of SURVEY Appendix A].
    volume_renderer.py:50-61; fields.py:49-73; gradient replaces fields.py:104-122."""
    n = x.shape[0]
    L = n_sdf_layers(sd)
    a = x
    cs = []
    for l in range(L):
        p = f"pts_linears.{l}."
        g, b = film_params(sd, p, w)
        g, b = _per_point(g, n), _per_point(b, n)
        u = a @ sd[p + "weight"].t() + sd[p + "bias"]
        phi = g * u + b
        a = torch.sin(phi)
        if want_grad:
            cs.append(g * torch.cos(phi))
    sdf = a @ sd["sigma_linear.weight"].t() + sd["sigma_linear.bias"]
    if not want_grad:
        return sdf, a
    gvec = sd["sigma_linear.weight"].expand(n, -1)
    for l in reversed(range(L)):
        gvec = (gvec * cs[l]) @ sd[f"pts_linears.{l}.weight"]
    return sdf, a, gvec


def sdf_gradient_autograd(sd, x, w, create_graph=False):
    """d sdf/dx through autograd exactly as the reference does it (fields.py:104-122)."""
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_forward(sd, x, w)[0].squeeze(-1)
        (gr,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph,
                                    retain_graph=create_graph)
    return gr


def color_head(csd, feat, grad, w):
    """sigmoid(rgb_linear(FiLM(cat[feat, grad]))); the raw (un-normalised) gradient is fed.
    src/models/fields.py:89-101."""
    n = feat.shape[0]
    g, b = film_params(csd, "views_linears.", w)
    g, b = _per_point(g, n), _per_point(b, n)
    u = torch.cat([feat, grad], -1) @ csd["views_linears.weight"].t() + csd["views_linears.bias"]
    h = torch.sin(g * u + b)
    return torch.sigmoid(h @ csd["rgb_linear.weight"].t() + csd["rgb_linear.bias"])


def inv_s_from_variance(variance):
    """clip(exp(10*variance), 1e-6, 1e6).  neus/models/fields.py:267-268; renderer.py:266."""
    return torch.exp(variance * 10.0).clamp(1e-6, 1e6)


# --------------------------------------------------------------------------------------
# a13/a14: rays
# --------------------------------------------------------------------------------------


def near_far_from_sphere(rays_o, rays_d):
    """src/models/generator.py:336-342."""
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def camera_matrices(cam_dist, fov_deg, resolution, dtype=torch.float32):
    """Pinhole intrinsics (4x4) + identity scene pose.  src/models/camera_network.py:9-28,
    src/utils/pose.py:190-206 (look_at((0,0,-1)) is the identity rotation)."""
    focal = (resolution / 2) * 1 / np.tan(0.5 * fov_deg * np.pi / 180.0)
    K = torch.tensor([[focal, 0, 0.5 * resolution, 0], [0, focal, 0.5 * resolution, 0],
                      [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
    K_inv = torch.tensor(np.linalg.inv(K.numpy()), dtype=torch.float32)
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([0.0, 0.0, -1.0]) * cam_dist
    w2c = invert_rot_t(c2w)
    return K.to(dtype), K_inv.to(dtype), c2w.to(dtype), w2c.to(dtype)


def invert_rot_t(pose):
    """src/utils/pose.py:143-154."""
    R = pose[..., :3, :3].transpose(-1, -2)
    t = -(R @ pose[..., :3, 3:4])
    out = torch.zeros_like(pose)
    out[..., :3, :3] = R
    out[..., :3, 3:4] = t
    out[..., 3, 3] = 1.0
    return out


def gen_rays(b2w, K_inv, c2w, w2c, cam_dist, resolution, scene_resolution):
    """Rays of an off-centre crop around the object.  generator.py:65-78 (c2b), 255-279, 317-333.
    Returns rays_o, rays_d (B,H,W,3), c2b, w2b."""
    w2b = invert_rot_t(b2w)
    c2b = w2b @ c2w
    b2c = w2c @ b2w
    t = b2c[:, :3, 3]
    R = resolution
    cx = cam_dist / t[:, 2] * t[:, 0] * R / 2 + 0.5 * scene_resolution
    cy = cam_dist / t[:, 2] * t[:, 1] * R / 2 + 0.5 * scene_resolution
    xo, yo = cx - R / 2, cy - R / 2
    lin = torch.linspace(0, 1, R, dtype=b2w.dtype)
    px = lin[None, None, :] * R + xo[:, None, None]          # (B, 1, W): varies along W
    py = lin[None, :, None] * R + yo[:, None, None]          # (B, H, 1)
    px, py = px.expand(-1, R, R), py.expand(-1, R, R)
    p = torch.stack([px, py, torch.ones_like(px)], -1)       # (B,H,W,3)
    p = p @ K_inv[:3, :3].t()
    v = p / torch.linalg.norm(p, dim=-1, keepdim=True)
    rays_d = torch.einsum("bij,bhwj->bhwi", c2b[:, :3, :3], v)
    rays_o = c2b[:, None, None, :3, 3].expand_as(rays_d)
    return rays_o, rays_d, c2b, w2b


# --------------------------------------------------------------------------------------
# a8-a11: coarse sampling, up-sampling, inverse-CDF, merge
# --------------------------------------------------------------------------------------


def coarse_z(near, far, S, jitter=None):
    """z_i = near + (far-near) i/(S-1); training jitter (u-0.5)*2/S per ray.
    renderer.py:359-373."""
    lin = torch.linspace(0.0, 1.0, S, dtype=near.dtype)
    z = near + (far - near) * lin[None, :]
    if jitter is not None:
        z = z + (jitter - 0.5) * 2.0 / S
    return z


def section_alpha(sdf_a, sdf_b, cos_v, dist, inv_s):
    prev_cdf = torch.sigmoid((0.5 * (sdf_a + sdf_b) - cos_v * dist * 0.5) * inv_s)
    next_cdf = torch.sigmoid((0.5 * (sdf_a + sdf_b) + cos_v * dist * 0.5) * inv_s)
    return (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)


def transmittance_weights(alpha):
    """w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-7).  renderer.py:174-175, 300."""
    ones = torch.ones_like(alpha[:, :1])
    T = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return alpha * T


def up_sample_weights(rays_o, rays_d, z, sdf, inv_s):
    """Section weights used for importance sampling.  renderer.py:137-178."""
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    radius = torch.linalg.norm(pts, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    pz, nz = z[:, :-1], z[:, 1:]
    cos_v = (ns - ps) / (nz - pz + 1e-5)
    prev_cos = torch.cat([torch.zeros_like(cos_v[:, :1]), cos_v[:, :-1]], -1)
    cos_v = torch.minimum(prev_cos, cos_v).clamp(-1e3, 0.0) * inside
    alpha = section_alpha(ps, ns, cos_v, nz - pz, inv_s)
    return transmittance_weights(alpha)


def sample_pdf_det(bins, weights, n):
    """Deterministic inverse-CDF sampling.  renderer.py:44-74 (det=True)."""
    wts = weights + 1e-5
    pdf = wts / wts.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, n, dtype=bins.dtype).expand(bins.shape[0], n).contiguous()
    ind = torch.searchsorted(cdf, u, right=True)
    below = (ind - 1).clamp(min=0)
    above = ind.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def merge_sorted(z, z_new, sdf=None, sdf_new=None):
    """cat + ascending sort (the sdf, when given, is permuted alongside).  renderer.py:183-197."""
    zc, idx = torch.sort(torch.cat([z, z_new], -1), dim=-1)
    if sdf is None:
        return zc, None
    return zc, torch.gather(torch.cat([sdf, sdf_new], -1), 1, idx)


def hierarchical_z(sd, rays_o, rays_d, near, far, w, S, I, K, jitter=None):
    """Coarse z -> K x (up_sample, merge); returns the merged (N, S+I) z.  renderer.py:359-413."""
    z = coarse_z(near, far, S, jitter)
    if I <= 0:
        return z
    N = z.shape[0]
    with torch.no_grad():
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
        sdf = sdf_forward(sd, pts.reshape(-1, 3), w)[0].reshape(N, S)
        for i in range(K):
            wts = up_sample_weights(rays_o, rays_d, z, sdf, 64.0 * 2 ** i)
            z_new = sample_pdf_det(z, wts, I // K)
            if i + 1 == K:
                z, _ = merge_sorted(z, z_new)
            else:
                p = rays_o[:, None, :] + rays_d[:, None, :] * z_new[..., None]
                s_new = sdf_forward(sd, p.reshape(-1, 3), w)[0].reshape(N, -1)
                z, sdf = merge_sorted(z, z_new, sdf, s_new)
    return z


# --------------------------------------------------------------------------------------
# a12: render_core on the merged z
# --------------------------------------------------------------------------------------


def render_core(sd, csd, variance, rays_o, rays_d, z, w, S, cos_anneal_ratio,
                grad_mode="analytic"):
    """NeuS compositing at section mid-points.  renderer.py:199-349, 448-473.
    Returns the dict `NeuSRenderer.render` returns (same keys/shapes)."""
    N, T = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 2.0 / S)], -1)
    mid_z = z + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]
    dirs = rays_d[:, None, :].expand(pts.shape).reshape(-1, 3)
    p = pts.reshape(-1, 3)
    if grad_mode == "analytic":
        sdf, feat, grad = sdf_forward(sd, p, w, want_grad=True)
    else:
        sdf, feat = sdf_forward(sd, p, w)
        grad = sdf_gradient_autograd(sd, p, w, create_graph=torch.is_grad_enabled())
    rgb = color_head(csd, feat, grad, w).reshape(N, T, 3)
    inv_s = inv_s_from_variance(variance)
    true_cos = (dirs * grad).sum(-1, keepdim=True)
    r = cos_anneal_ratio
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - r) + F.relu(-true_cos) * r)
    d = dists.reshape(-1, 1)
    prev_cdf = torch.sigmoid((sdf - iter_cos * d * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf + iter_cos * d * 0.5) * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(N, T).clamp(0.0, 1.0)
    weights = transmittance_weights(alpha)
    pts_norm = torch.linalg.norm(p, dim=-1).reshape(N, T)
    relax = (pts_norm < 1.2).to(z.dtype)
    gnorm = torch.linalg.norm(grad.reshape(N, T, 3), dim=-1)
    eik = (relax * (gnorm - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    return {
        "s_val": (1.0 / inv_s).expand(N, 1),
        "cdf_fine": prev_cdf.reshape(N, T),
        "weight_sum": weights.sum(-1, keepdim=True),
        "weight_max": weights.max(-1, keepdim=True)[0],
        "gradients": grad.reshape(N, T, 3),
        "weights": weights,
        "gradient_error": eik,
        "inside_sphere": (pts_norm < 1.0).to(z.dtype),
        "mid_z_vals": mid_z,
        "surface_loss": torch.exp(-1e2 * sdf.abs()).mean(),
        "sdf": sdf.reshape(N, T),
        "pts_norm": pts_norm,
        "pts": pts,
        "color_fine": (rgb * weights[..., None]).sum(1),
        "raw_color": rgb,
        "alpha": alpha,
    }


def render(sd, csd, variance, rays_o, rays_d, near, far, w, S, I, K, cos_anneal_ratio,
           jitter=None, grad_mode="analytic"):
    """`NeuSRenderer.render` (n_outside = 0).  renderer.py:351-473."""
    z = hierarchical_z(sd, rays_o, rays_d, near, far, w, S, I, K, jitter)
    return render_core(sd, csd, variance, rays_o, rays_d, z, w, S, cos_anneal_ratio, grad_mode)


# --------------------------------------------------------------------------------------
# a15: Phong shading + maps
# --------------------------------------------------------------------------------------


def light_terms(lsd, w2b):
    """Directional light in the box frame + scalar colours.
    src/models/lighting.py:33-52, 115-119."""
    d = lsd["param_direction"]
    d = d / torch.linalg.norm(d)
    dirs = torch.einsum("bij,j->bi", w2b[:, :3, :3], d)
    amb = torch.sigmoid(lsd["param_ambient"])
    return dirs, amb, 1.0 - amb, lsd["param_specular"].clamp(min=0), lsd["param_shininess"]


def render_maps(ro, rays_o, lsd, w2b, bg, B, H, W_, return_raw=False):
    """Per-sample Phong shading and weighted sums to image-space maps.
    src/models/generator.py:80-174; lighting.py:126-170, 173-225.
    `ro` = render dict; rays_o (N,3); bg (B,3)."""
    N, T, _ = ro["pts"].shape
    wts = ro["weights"][..., None]
    ldir, amb, cd, cs, sh = light_terms(lsd, w2b)
    ldir = F.normalize(ldir, dim=-1, eps=1e-6)
    l = _per_point(ldir, N)[:, None, :]
    n = F.normalize(ro["gradients"], dim=-1, eps=1e-6)
    ndl = (n * l).sum(-1, keepdim=True)
    diff = cd * F.relu(ndl)
    shade = (amb + diff).expand(N, T, 3)
    view = F.normalize(rays_o[:, None, :] - ro["pts"], dim=-1, eps=1e-6)
    refl = -l + 2.0 * ndl * n
    al = F.relu((view * refl).sum(-1, keepdim=True)) * (ndl > 0).to(wts.dtype)
    spec = (cs * torch.pow(al, sh)).expand(N, T, 3)

    def to_map(x):
        return x.reshape(B, H, W_, -1).permute(0, 3, 1, 2)

    def wsum(x):
        return to_map((x * wts).sum(1))

    wsum_map = to_map(ro["weight_sum"])
    no_spec = wsum(shade * ro["raw_color"])
    spec_map = wsum(spec)
    rgb = no_spec + spec_map
    bgm = bg[:, :, None, None].expand(B, 3, H, W_)
    out = {
        "weight_sum_map": wsum_map,
        "color_map": to_map(ro["color_fine"]),
        "shading_map": wsum(shade),
        "image_no_bg": rgb,
        "image": rgb + bgm * (1 - wsum_map),
        "mask": wsum_map.clamp(1e-3, 1.0 - 1e-3),
    }
    if return_raw:
        out.update({
            "amb_shading_map": wsum(amb.expand(N, T, 3)),
            "diff_shading_map": wsum(diff.expand(N, T, 3)),
            "normal_map": wsum(ro["gradients"]),
            "no_specular_map": no_spec,
            "specular_map": spec_map,
            "z_map": to_map((ro["mid_z_vals"] * ro["weights"]).sum(-1, keepdim=True)),
            "z_min": ro["mid_z_vals"].min(-1).values.reshape(B, -1).min(-1).values,
        })
    return out


# --------------------------------------------------------------------------------------
# a16-a19: discriminator, upfirdn2d, ADA geometric augmentation
# --------------------------------------------------------------------------------------

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]


def hz_geom(dtype=torch.float32):
    """Normalised sym6 low-pass (12 taps).  ada/augment.py:169; upfirdn2d.py setup_filter."""
    f = torch.tensor(SYM6, dtype=torch.float32)
    return (f / f.sum()).to(dtype)


def upfirdn2d(x, f, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0), flip=False, gain=1.0):
    """zero-insert upsample -> pad/crop -> FIR -> decimate, 2-D filter f (fh, fw).
    ada/torch_utils/ops/upfirdn2d.py:168-208 (`_upfirdn2d_ref`); up/down/pad as (x, y) /
    (x0, x1, y0, y1)."""
    Bn, C, Hh, Ww = x.shape
    ux, uy = up
    dx, dy = down
    px0, px1, py0, py1 = pad
    y = x.reshape(Bn, C, Hh, 1, Ww, 1)
    y = F.pad(y, [0, ux - 1, 0, 0, 0, uy - 1]).reshape(Bn, C, Hh * uy, Ww * ux)
    y = F.pad(y, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    y = y[:, :, max(-py0, 0): y.shape[2] - max(-py1, 0), max(-px0, 0): y.shape[3] - max(-px1, 0)]
    k = (f * gain).to(x.dtype)
    if not flip:
        k = k.flip([0, 1])
    k = k[None, None].repeat(C, 1, 1, 1)
    y = F.conv2d(y, k, groups=C)
    return y[:, :, ::dy, ::dx]


def upfirdn2d_separable(x, f1, up=1, down=1, pad=(0, 0, 0, 0), flip=False, gain=1.0):
    """Separable f1 (taps,) applied along x then y, as the plugin path does
    (upfirdn2d.py:239-241: two passes with sqrt(gain) each)."""
    g = math.sqrt(gain)
    y = upfirdn2d(x, f1[None, :], (up, 1), (down, 1), (pad[0], pad[1], 0, 0), flip, g)
    return upfirdn2d(y, f1[:, None], (1, up), (1, down), (0, 0, pad[2], pad[3]), flip, g)


def upsample2d(x, f1, up=2):
    """upfirdn2d.py:325-357: padding from filter size, gain = up^2."""
    fw = f1.numel()
    p = [(fw + up - 1) // 2, (fw - up) // 2]
    return upfirdn2d_separable(x, f1, up=up, pad=(p[0], p[1], p[0], p[1]), gain=float(up * up))


def downsample2d(x, f1, down=2, padding=0, flip=False):
    """upfirdn2d.py:360-392."""
    fw = f1.numel()
    p = [padding + (fw - down + 1) // 2, padding + (fw - down) // 2]
    return upfirdn2d_separable(x, f1, down=down, pad=(p[0], p[1], p[0], p[1]), flip=flip)


def ada_G_inv(B, width, height, t_frac, s, dtype=torch.float32):
    """G_inv = T(-round(t*[W,H])) @ S(1/s) for the xint + scale branches.
    ada/augment.py:213-230.  t_frac (B,2) in [-xint_max, xint_max], s (B,)."""
    G = torch.eye(3, dtype=dtype).repeat(B, 1, 1)
    T_ = torch.eye(3, dtype=dtype).repeat(B, 1, 1)
    T_[:, 0, 2] = -torch.round(t_frac[:, 0] * width)
    T_[:, 1, 2] = -torch.round(t_frac[:, 1] * height)
    S_ = torch.eye(3, dtype=dtype).repeat(B, 1, 1)
    S_[:, 0, 0] = 1.0 / s
    S_[:, 1, 1] = 1.0 / s
    return G @ T_ @ S_


def _t2d(tx, ty, dtype):
    m = torch.eye(3, dtype=dtype)
    m[0, 2], m[1, 2] = tx, ty
    return m


def _s2d(sx, sy, dtype):
    m = torch.eye(3, dtype=dtype)
    m[0, 0], m[1, 1] = sx, sy
    return m


def ada_margins(G_inv, width, height, hz_pad=3):
    """Reflect-pad margins (mx0, my0, mx1, my1) as python ints.  augment.py:272-283."""
    cx, cy = (width - 1) / 2, (height - 1) / 2
    cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=G_inv.dtype)
    cp = G_inv @ cp.t()
    m = cp[:, :2, :].permute(1, 0, 2).flatten(1)
    m = torch.cat([-m, m]).max(dim=1).values
    m = m + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=G_inv.dtype)
    m = m.clamp(min=0)
    m = torch.minimum(m, torch.tensor([width - 1, height - 1] * 2, dtype=G_inv.dtype))
    return [int(v) for v in m.ceil().to(torch.int32)]


def affine_bilinear_sample(x, theta, Ho, Wo):
    """F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=False) restated with gathers so that
    it is differentiable to any order (ATen's grid_sampler backward has no derivative formula, which
    is why the reference carries grid_sample_gradfix.py:69-97)."""
    B, C, Hi, Wi = x.shape
    dt = x.dtype
    xs = (2.0 * torch.arange(Wo, dtype=dt) + 1.0) / Wo - 1.0
    ys = (2.0 * torch.arange(Ho, dtype=dt) + 1.0) / Ho - 1.0
    gx = theta[:, 0, 0, None, None] * xs[None, None, :] + theta[:, 0, 1, None, None] * ys[None, :, None] + theta[:, 0, 2, None, None]
    gy = theta[:, 1, 0, None, None] * xs[None, None, :] + theta[:, 1, 1, None, None] * ys[None, :, None] + theta[:, 1, 2, None, None]
    ix = ((gx + 1.0) * Wi - 1.0) * 0.5
    iy = ((gy + 1.0) * Hi - 1.0) * 0.5
    x0, y0 = torch.floor(ix), torch.floor(iy)
    tx, ty = ix - x0, iy - y0
    x0, y0 = x0.long(), y0.long()
    out = 0
    flat = x.reshape(B, C, Hi * Wi)
    for dy, wy in ((0, 1 - ty), (1, ty)):
        for dx, wx in ((0, 1 - tx), (1, tx)):
            xi, yi = x0 + dx, y0 + dy
            ok = ((xi >= 0) & (xi < Wi) & (yi >= 0) & (yi < Hi)).to(dt)
            idx = (yi.clamp(0, Hi - 1) * Wi + xi.clamp(0, Wi - 1)).reshape(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
            v = torch.gather(flat, 2, idx).reshape(B, C, Ho, Wo)
            out = out + v * (wx * wy * ok)[:, None]
    return out


def ada_geometric(images, G_inv):
    """Execute the geometric part of AugmentPipe for a given per-sample G_inv (B,3,3):
    reflect pad -> x2 sym6 upsample -> affine bilinear resample -> /2 sym6 downsample.
    ada/augment.py:270-301.  Returns (out, intermediates)."""
    B, C, H, W_ = images.shape
    dt = images.dtype
    f1 = hz_geom(dt)
    hz_pad = 3
    mx0, my0, mx1, my1 = ada_margins(G_inv, W_, H, hz_pad)
    x = F.pad(images, [mx0, mx1, my0, my1], mode="reflect")
    G = _t2d((mx0 - mx1) / 2, (my0 - my1) / 2, dt) @ G_inv
    up = upsample2d(x, f1, up=2)
    G = _s2d(2, 2, dt) @ G @ _s2d(0.5, 0.5, dt)
    G = _t2d(-0.5, -0.5, dt) @ G @ _t2d(0.5, 0.5, dt)
    shape = [B, C, (H + hz_pad * 2) * 2, (W_ + hz_pad * 2) * 2]
    G = _s2d(2 / up.shape[3], 2 / up.shape[2], dt) @ G @ _s2d(shape[3] / 2, shape[2] / 2, dt)
    theta = G[:, :2, :]
    smp = affine_bilinear_sample(up, theta, shape[2], shape[3])
    out = downsample2d(smp, f1, down=2, padding=-hz_pad * 2, flip=True)
    return out, {"padded": x, "up": up, "theta": theta, "sampled": smp,
                 "margins": (mx0, my0, mx1, my1)}


def dc_discriminator(dsd, x):
    """[conv4x4 s2 p1 (no bias) -> lrelu 0.2] x n, then conv4x4 valid.
    src/models/discriminator.py:57-85."""
    n = len([k for k in dsd if k.startswith("blocks.")])
    for i in range(n):
        x = F.leaky_relu(F.conv2d(x, dsd[f"blocks.{i}.weight"], stride=2, padding=1), 0.2)
    out = F.conv2d(x, dsd["conv_out.weight"], dsd.get("conv_out.bias"))
    return out.reshape(x.shape[0], -1)


def seeded_conv_weights(shapes, seed):
    """Test input recipe (NOT reference arithmetic): one seeded uniform(-b, b) draw per tensor with b = sqrt(6 / (1.04 fan_in))
    -- the variance-preserving bound for LeakyReLU(0.2) layers, so that the logits of the six-layer network of
    discriminator.py:63-72 are O(1) and a forward tolerance means something (nn.Conv2d's default bound 1/sqrt(fan_in) shrinks
    the signal 2.4x per layer: logits of 1e-3).  `shapes`: {name: shape} in state_dict order.  The 128 x 128 fixture (tests/golden/f14_discriminator_128.npz) stores check sums of these tensors
    instead of 2 x 11.4 MB of weights; both the generating script and the test call this function."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(int(seed) * 1000 + i)
        fan_in = 1
        for d in shp[1:]:
            fan_in *= int(d)
        b = math.sqrt(6.0 / (1.04 * max(1, fan_in)))
        out[k] = (torch.rand(*shp, generator=g) * 2.0 - 1.0) * b
    return out


# --------------------------------------------------------------------------------------
# a20: losses
# --------------------------------------------------------------------------------------


def bce_logits_const(d_out, target):
    """src/loss/gan.py:19-22."""
    return F.binary_cross_entropy_with_logits(d_out, torch.full_like(d_out, float(target)))


def r1_penalty(d_out, x_in):
    """mean_b sum (d sum(D)/dx)^2.  src/loss/gan.py:5-14."""
    (g,) = torch.autograd.grad(d_out.sum(), x_in, create_graph=True, retain_graph=True)
    return g.pow(2).reshape(x_in.shape[0], -1).sum(1).mean()


def pose_to_vec(pose):
    """src/utils/pose_sampler.py:20-22."""
    return pose[..., :2, :3].flatten(-2, -1)
