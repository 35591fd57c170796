"""Autograd structure over the HIP kernels (one torch.autograd.Function per differentiable op).

Forward kernels: csrc/mlp.hip, csrc/render.hip, csrc/disc.hip.  Backward kernels:
csrc/*_bwd.hip.  An op whose backward kernel is not available raises instead of silently falling
back to anything else."""
import torch

from . import ops


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ------------------------------------------------------------------------------------------
# a1/a2: style MLP + FiLM parameters
# ------------------------------------------------------------------------------------------

def film_params(pack, z=None, w=None):
    """-> (w, gamma[B,9,128], beta[B,9,128]).  Forward in one HIP launch.  These are O(B*128*64)
    flops; when parameter gradients are required the (tiny) graph is rebuilt with torch ops on the
    same device so that autograd reaches the reference-named parameters."""
    src = z if z is not None else w
    if _needs_grad(src, *pack.param_lists()[1]):
        P = pack.film_stacked(differentiable=True)
        return FilmParamsFunction.apply(z, w, P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"])
    P = pack.film_stacked(differentiable=False)
    with torch.no_grad():
        return ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], z=z, w=w)


class FilmParamsFunction(torch.autograd.Function):
    """oi_film_params / oi_film_params_bwd.  First-order only (the FiLM heads are linear in their parameters and the
    training losses never differentiate a gradient with respect to them a second time)."""

    @staticmethod
    def forward(ctx, z, w, style_w, style_b, gw, gb, bw, bb):
        w_out, gamma, beta = ops.film_params(style_w, style_b, gw, gb, bw, bb, z=z, w=w)
        ctx.save_for_backward(z, w_out, style_w, style_b, gw, bw)
        ctx.from_z = z is not None
        ctx.set_materialize_grads(False)   # (backward handles the absent ones)
        ctx.need_dz = bool(z is not None and z.requires_grad)
        return w_out, gamma, beta

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_w_out, d_gamma, d_beta):
        z, w_out, style_w, style_b, gw, bw = ctx.saved_tensors
        if d_gamma is None:
            d_gamma = torch.zeros(w_out.shape[0], gw.shape[0], 128, device=w_out.device)
        if d_beta is None:
            d_beta = torch.zeros_like(d_gamma)
        r = ops.film_params_bwd(d_gamma, d_beta, w_out, gw, bw, style_w if ctx.from_z else None,
                                style_b if ctx.from_z else None, z if ctx.from_z else None, d_w_in=d_w_out,
                                want_dz=ctx.need_dz)
        if ctx.from_z:
            return r.get("d_z"), None, r["d_style_w"], r["d_style_b"], r["d_gw"], r["d_gb"], r["d_bw"], r["d_bb"]
        return None, r["d_w"], None, None, r["d_gw"], r["d_gb"], r["d_bw"], r["d_bb"]


class StyleFunction(torch.autograd.Function):
    """The style MLP alone (`ShapeNetwork.style(z)`, fields.py:15-21; generator.py:237 calls it under autograd): the
    forward is oi_film_params without FiLM layers, the backward oi_film_params_bwd with NL = 0 (style_bwd_kernel)."""

    @staticmethod
    def forward(ctx, z, ws, bs):
        w = ops.film_params(ws, bs, None, None, None, None, z=z)[0]
        ctx.save_for_backward(z, ws, bs)
        ctx.need_dz = bool(z.requires_grad)
        return w

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_w):
        z, ws, bs = ctx.saved_tensors
        r = ops.style_bwd(d_w, ws, bs, z, want_dz=ctx.need_dz)
        return r.get("d_z"), r["d_style_w"], r["d_style_b"]


def style_mlp(style_module, z):
    ws = torch.stack([m.weight for m in style_module])
    bs = torch.stack([m.bias for m in style_module])
    if _needs_grad(z, ws, bs):
        return StyleFunction.apply(z, ws, bs)
    with torch.no_grad():
        return ops.film_params(ws, bs, None, None, None, None, z=z)[0]


# ------------------------------------------------------------------------------------------
# a3-a6: the MLP
# ------------------------------------------------------------------------------------------

def sdf_mlp(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat, scratch=None, blob_ready=False):
    """-> (sdf (n,), grad (n,3)|None, rgb (n,3)|None, feat (n,128)|None).  scratch / blob_ready: ops.sdf_mlp_fwd (no-grad path)."""
    if _needs_grad(pts, gamma, beta, *pack.param_lists()[0]):
        from .autograd_mlp import SdfMlpFunction
        return SdfMlpFunction.run(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat)
    with torch.no_grad():
        sdf, grad, rgb, feat, _ = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig,
                                                  want_grad, want_rgb, want_feat, scratch, blob_ready)
    return sdf, grad, rgb, feat


class ColorHeadFunction(torch.autograd.Function):
    """ColorNetwork.forward on caller-supplied features (fields.py:89-101): oi_color_head_fwd / oi_color_head_bwd.  First
    order: the head is a plain function of (features, normals, FiLM rows, weights) -- the second-order terms of a training loss
    live in whatever produced `normals` (ShapeNetwork.gradient), not here."""

    @staticmethod
    def forward(ctx, feat, normals, gamma, beta, wv, bv, wrgb, brgb, B):
        args = [ops._c(t) for t in (feat, normals, gamma, beta, wv, bv, wrgb, brgb)]
        ctx.save_for_backward(*args)
        ctx.B = B
        return ops.color_head_fwd(*args, B)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb):
        return ops.color_head_bwd(*ctx.saved_tensors, ops._c(g_rgb), ctx.B) + (None,)


def color_head(feat, normals, gamma, beta, wv, bv, wrgb, brgb, B):
    if _needs_grad(feat, normals, gamma, beta, wv, bv, wrgb, brgb):
        return ColorHeadFunction.apply(feat, normals, gamma, beta, wv, bv, wrgb, brgb, B)
    with torch.no_grad():
        return ops.color_head_fwd(*[ops._c(t) for t in (feat, normals, gamma, beta, wv, bv, wrgb, brgb)], B)


# ------------------------------------------------------------------------------------------
# a12/a15: compositing + shading maps
# ------------------------------------------------------------------------------------------

def composite(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light, cos_anneal_ratio, B,
              outputs=None, image_planar=False):
    """`image_planar`: 'image' comes back as (B, 3, N / B) -- the (B, 3, H, W) map itself -- and its gradient is read in that
    form.  With a gradient recorded the dict also carries 'finals' / 'ray_sums' (no gradient: logging scalars)."""
    if _needs_grad(sdf, grad, rgb, variance, light, light_dir):
        from .autograd_render import CompositeFunction
        return CompositeFunction.run(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light,
                                     cos_anneal_ratio, B, outputs, image_planar)
    with torch.no_grad():
        return ops.composite_fwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light,
                                 cos_anneal_ratio, B, outputs, image_planar=image_planar)
