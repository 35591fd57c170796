"""GAN losses of the path (host-side glue over D outputs): src/loss/gan.py:5-22, 39-49 and
src/loss/position.py:4-18.  The R1 term differentiates through oi_amd's discriminator, whose
autograd Functions supply the double-backward from HIP kernels."""
import torch
import torch.nn.functional as F
from torch import autograd


def compute_grad2(d_out, x_in):
    batch_size = x_in.size(0)
    (grad_dout,) = autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                                 only_inputs=True)
    return grad_dout.pow(2).reshape(batch_size, -1).sum(1).mean()


class GANLoss:
    def __init__(self, gan_str):
        if gan_str != "bce":
            raise NotImplementedError(gan_str)

    def __call__(self, d_out, target):
        assert d_out.dim() == 2 and d_out.shape[1] == 1, d_out.shape
        return F.binary_cross_entropy_with_logits(d_out, d_out.new_full(d_out.size(), float(target)))


class PositionLoss:
    def __init__(self, loss_str):
        self.loss = {"mse": F.mse_loss, "smooth_l1": F.smooth_l1_loss}[loss_str]

    def __call__(self, pred, target, reduction="mean"):
        return self.loss(pred, target, reduction=reduction)


class _GanLosses(autograd.Function):
    """The scalar losses of one step in ONE launch each way (csrc/loss.hip).  At batch 1 every ATen op of the composition
    above is a ~4 us launch on 1..7 elements: 45 of the ~200 launches of a discriminator step."""

    @staticmethod
    def forward(ctx, d_real, d_fake, pose, gx, aux_w, reg_w):
        from . import ops
        ts = [None if t is None else t.detach().contiguous() for t in (d_real, d_fake, pose, gx, aux_w)]
        out = ops.gan_losses_fwd(*ts, reg_w)
        ctx.ts, ctx.reg_w = ts, reg_w
        ctx.set_materialize_grads(False)
        total, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        from . import ops
        d_real, d_fake, pose, gx, aux_w = ctx.ts
        if g_total is None:
            return (None,) * 6
        need = ctx.needs_input_grad
        g = ops.gan_losses_bwd(g_total.contiguous(), d_real, d_fake, pose, gx, aux_w, ctx.reg_w,
                               d_real is not None and need[0], d_fake is not None and need[1], gx is not None and need[3])
        return g[0], g[1], None, g[2], None, None


class _WeightedSum(autograd.Function):
    """sum_i w_i * term_i over device scalars in one launch each way (oi_weighted_sum_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        from . import ops
        ctx.weights, ctx.shapes = weights, [t.shape for t in terms]
        ctx.set_materialize_grads(False)
        return ops.weighted_sum_fwd([t.detach().reshape(()) for t in terms], weights)

    @staticmethod
    def backward(ctx, g_out):
        if g_out is None:
            return (None,) * (1 + len(ctx.weights))
        from . import ops
        g = ops.weighted_sum_bwd(g_out.contiguous(), ctx.weights, g_out.device)
        return (None,) + tuple(g[i].view(sh) if ctx.needs_input_grad[1 + i] else None for i, sh in enumerate(ctx.shapes))


def weighted_sum(terms, weights):
    """sum_i weights[i] * terms[i] for 1..8 scalar float32 loss terms on the device (gan_pose_trainer.py:122-137): one
    launch each way.  (No tensor-op fallback: like every op of the path it raises for host tensors.)"""
    if not 1 <= len(terms) <= 8 or any(t.numel() != 1 for t in terms):
        raise ValueError(f"weighted_sum: {len(terms)} terms of sizes {[t.numel() for t in terms]} (1..8 scalars)")
    return _WeightedSum.apply(tuple(float(w) for w in weights), *terms)


class _GanLossesCat(autograd.Function):
    """The same for ONE discriminator pass over [real batch; fake batch] (oi_amd.graphed.GraphedDStep): d_all [2B, K], gx_all
    [2B, ...] = d sum(d_all[:B, 0]) / d x_all (its fake half is exactly zero), one gradient tensor per input."""

    @staticmethod
    def forward(ctx, d_all, pose, gx_all, aux_w, reg_w, B):
        from . import ops
        d_all, gx_all = d_all.detach().contiguous(), gx_all.detach().contiguous()
        pose = None if pose is None else pose.detach().contiguous()
        out = ops.gan_losses_fwd(d_all[:B], d_all[B:], pose, gx_all.view(B, -1), aux_w, reg_w)
        ctx.ts, ctx.reg_w, ctx.B = (d_all, pose, gx_all, aux_w), reg_w, B
        ctx.set_materialize_grads(False)
        total, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        from . import ops
        d_all, pose, gx_all, aux_w = ctx.ts
        if g_total is None:
            return (None,) * 6
        B = ctx.B
        g_all, g_gx = torch.empty_like(d_all), torch.empty_like(gx_all)
        ops.gan_losses_bwd(g_total.contiguous(), d_all[:B], d_all[B:], pose, gx_all.view(B, -1), aux_w, ctx.reg_w, True, True,
                           True, out=(g_all[:B], g_all[B:], g_gx.view(B, -1)))
        return g_all, None, g_gx, None, None, None


def gan_losses_cat(d_all, n_real, pose=None, gx_all=None, aux_w=None, reg_w=0.0):
    return _GanLossesCat.apply(d_all, pose, gx_all, aux_w, float(reg_w), int(n_real))


def gan_losses(d_real=None, d_fake=None, pose=None, gx=None, aux_w=None, reg_w=0.0):
    """-> (total, parts) with total = BCE(d_real[:, :1], 1) + BCE(d_fake[:, :1], 0) + reg_w * R1(gx) + aux_w * MSE(d_fake[:, 1:],
    pose) (absent terms: None) and parts = [real + fake, reg, fake, real, aux] (no gradient).  `aux_w`: device scalar tensor.
    Same arithmetic as GANLoss / compute_grad2 / PositionLoss above, which stay the reference-facing interface."""
    return _GanLosses.apply(d_real, d_fake, pose, gx, aux_w, float(reg_w))


_ONES = {}


def ones_scalar(like):
    """A cached tensor of ones shaped like `like` (the seed of a backward pass: autograd's default is a fill per call)."""
    key = (tuple(like.shape), like.device)
    ones = _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones(like.shape, device=like.device)
    return ones


def grad_wrt_input(d_out, x_in, grad_outputs=None):
    """d sum(d_out) / d x_in with the graph kept (the R1 penalty's inner gradient, compute_grad2 above) -- `grad_outputs`
    from a cached tensor of ones instead of a `sum()` whose backward expands one (or the caller's selection)."""
    key = (tuple(d_out.shape), d_out.device)
    ones = grad_outputs if grad_outputs is not None else _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones(d_out.shape, device=d_out.device)
    from . import autograd_conv
    autograd_conv.INPUT_GRAD_ONLY += 1   # (the convolutions skip their weight gradients: nothing here asks for them)
    try:
        (g,) = autograd.grad(outputs=d_out, inputs=x_in, grad_outputs=ones, create_graph=True, retain_graph=True,
                             only_inputs=True)
    finally:
        autograd_conv.INPUT_GRAD_ONLY -= 1
    return g


def linear_increase(max_it, max_weight):
    return lambda it: min(it / max_it, 1) * max_weight
