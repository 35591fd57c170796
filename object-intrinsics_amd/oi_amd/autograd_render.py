"""autograd.Function of the fused compositing + Phong maps kernel (csrc/render.hip / render_bwd.hip)."""
import torch
from torch.autograd.function import once_differentiable

from . import ops

OUT_KEYS = ops.PER_SAMPLE_OUT + tuple(ops.PER_RAY_OUT) + ("reduce4", "ray_sums", "finals")
NON_DIFF = ("cdf", "alpha", "inside_sphere", "pts_norm", "weight_max", "ray_sums", "finals")


class CompositeFunction(torch.autograd.Function):
    @staticmethod
    def run(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light, cos_anneal_ratio, B,
            outputs=None, image_planar=False):
        # the kernel normalises the light direction; doing it here too (idempotent) lets autograd own the
        # Jacobian of the normalisation, the kernel returns d/d(unit vector)
        # (a direction from DirectionalLight.batch_direction_unit is already that, with its own backward)
        ldir_n = light_dir if getattr(light_dir, "_oi_unit", False) else torch.nn.functional.normalize(light_dir, dim=-1, eps=1e-6)
        outs = CompositeFunction.apply(sdf, grad, rgb, variance, light, ldir_n, dists, mid_z, rays_o, rays_d, bg,
                                       float(cos_anneal_ratio), B, bool(image_planar),
                                       None if outputs is None else tuple(outputs))
        # (outputs nobody asked for are not computed: null pointers for the kernel, None here; 'finals' / 'ray_sums' come with
        # 'reduce4', as from ops.composite_fwd)
        res = {k: v for k, v in zip(OUT_KEYS, outs) if v is not None}
        return res

    @staticmethod
    def forward(ctx, sdf, grad, rgb, variance, light, ldir_n, dists, mid_z, rays_o, rays_d, bg, car, B, planar, want):
        out = ops.composite_fwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, ldir_n, bg, variance, light, car, B,
                                outputs=want, image_planar=planar)
        # ~20 outputs of which a loss touches a few: absent upstream gradients arrive as None (a null pointer for the kernel),
        # not as one zero-filled tensor -- one fill launch -- each
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(sdf, grad, rgb, variance, light, ldir_n, dists, mid_z, rays_o, rays_d, bg)
        ctx.car, ctx.B, ctx.planar = car, B, planar
        ctx.mark_non_differentiable(*[out[k] for k in NON_DIFF if k in out])
        return tuple(out.get(k) for k in OUT_KEYS)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        sdf, grad, rgb, variance, light, ldir_n, dists, mid_z, rays_o, rays_d, bg = ctx.saved_tensors
        g = {k: v for k, v in zip(OUT_KEYS, gouts) if k in ops.GRAD_IN and v is not None}
        d_sdf, d_grad, d_rgb, d_var, d_light, d_ldir = ops.composite_bwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d,
                                                                         ldir_n, bg, variance, light, ctx.car, ctx.B, g,
                                                                         image_planar=ctx.planar)
        return (d_sdf, d_grad, d_rgb, d_var.reshape(variance.shape), d_light, d_ldir) + (None,) * 9
