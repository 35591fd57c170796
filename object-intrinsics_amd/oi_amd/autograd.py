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
        return _film_params_torch(pack.film_stacked(differentiable=True), z, w)
    P = pack.film_stacked(differentiable=False)
    with torch.no_grad():
        return ops.film_params(P["style_w"], P["style_b"], P["gw"], P["gb"], P["bw"], P["bb"], z=z, w=w)


def _film_params_torch(P, z, w):
    if w is None:
        h = z
        for i in range(3):
            h = torch.nn.functional.leaky_relu(h @ P["style_w"][i].t() + P["style_b"][i], 0.2)
        w = h
    gamma = 15.0 * (torch.einsum("bk,lfk->blf", w, P["gw"]) + P["gb"][None]) + 30.0
    beta = 0.25 * (torch.einsum("bk,lfk->blf", w, P["bw"]) + P["bb"][None])
    return w, gamma, beta


def style_mlp(style_module, z):
    ws = torch.stack([m.weight for m in style_module])
    bs = torch.stack([m.bias for m in style_module])
    if _needs_grad(z, ws, bs):
        h = z
        for i in range(3):
            h = torch.nn.functional.leaky_relu(h @ ws[i].t() + bs[i], 0.2)
        return h
    with torch.no_grad():
        return ops.film_params(ws, bs, None, None, None, None, z=z)[0]


# ------------------------------------------------------------------------------------------
# a3-a6: the MLP
# ------------------------------------------------------------------------------------------

def sdf_mlp(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat, scratch=None):
    """-> (sdf (n,), grad (n,3)|None, rgb (n,3)|None, feat (n,128)|None)."""
    if _needs_grad(pts, gamma, beta, *pack.param_lists()[0]):
        from .autograd_mlp import SdfMlpFunction
        return SdfMlpFunction.run(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat)
    with torch.no_grad():
        sdf, grad, rgb, feat, _ = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig,
                                                  want_grad, want_rgb, want_feat, scratch)
    return sdf, grad, rgb, feat


# ------------------------------------------------------------------------------------------
# a12/a15: compositing + shading maps
# ------------------------------------------------------------------------------------------

def composite(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light, cos_anneal_ratio, B,
              outputs=None):
    if _needs_grad(sdf, grad, rgb, variance, light, light_dir):
        from .autograd_render import CompositeFunction
        return CompositeFunction.run(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light,
                                     cos_anneal_ratio, B, outputs)
    with torch.no_grad():
        return ops.composite_fwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light,
                                 cos_anneal_ratio, B, outputs)
