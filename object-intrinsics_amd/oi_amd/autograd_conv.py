"""Autograd of the 4x4 convolutions: three Functions (forward conv, data gradient, weight gradient)
whose backward passes are expressed with each other, so derivatives of any order run on the HIP
implicit-GEMM kernels (csrc/disc.hip, csrc/disc_bwd.hip).  The fused LeakyReLU is differentiated
through the saved *output* (sign(y) == sign(pre-activation))."""
import os

import torch

from . import ops

# plain (not create_graph) backward passes: LeakyReLU mask applied on load by the gradient kernels, weight gradients added
# straight into a registered accumulator.  OI_CONV_FUSED_BWD=0: the staged ops (A/B switch)
FUSED_BWD = os.environ.get("OI_CONV_FUSED_BWD", "1") == "1"
MASK_FIRST = os.environ.get("OI_CONV_MASK_FIRST", "1") == "1"
# discriminator forward under autograd as a chain of pre-activations (_ConvPre).  OI_CONV_PRE=0: conv + activation per layer
PRE_CHAIN = os.environ.get("OI_CONV_PRE", "1") == "1"

# > 0 while a caller differentiates with respect to the network INPUT only (the R1 penalty's inner gradient,
# losses.grad_wrt_input): `ctx.needs_input_grad` is fixed at forward time and says "the weight requires grad", so every
# layer would compute a weight gradient that `autograd.grad(..., inputs=x)` throws away -- five launches per discriminator
# step, and five dead nodes in the double-backward graph
INPUT_GRAD_ONLY = 0


def _want_w(ctx):
    return ctx.needs_input_grad[1] and not INPUT_GRAD_ONLY


def _wgrad(w, g, x, stride, pad, mask_ref=None, slope=1.0):
    """Weight gradient of `w` inside a backward pass.  A plain backward (grad mode off: nothing will differentiate the
    result) with a registered accumulator (ops.GradSink) adds into it and returns None; otherwise the differentiable node."""
    if FUSED_BWD and not torch.is_grad_enabled():
        acc = ops.GradSink.lookup(w)
        if acc is not None:
            ops.conv4x4_wgrad(g, x, stride, pad, mask_ref, slope, acc=acc)
            return None
        return ops.conv4x4_wgrad(g, x, stride, pad, mask_ref, slope)
    if mask_ref is not None:
        g = _MaskMul.apply(g, mask_ref, slope)
    return _Wgrad.apply(g, x, stride, pad)


def _bwd_plain(ctx, w, gy, x, stride, pad, mask_ref=None, slope=1.0):
    """Both gradients of a layer in a plain backward: ONE launch when both are wanted (ops.conv4x4_bwd)."""
    want_x, want_w = ctx.needs_input_grad[0], _want_w(ctx)
    if mask_ref is not None and want_x and MASK_FIRST:
        # the data-gradient kernel reads an incoming-gradient value once per tap: a mask applied on load costs it more (31 vs
        # ~20 us per layer at two images) than the 5 us launch that writes the masked gradient out first.  A weight gradient
        # alone keeps the mask on load (one read per value).
        gy, mask_ref = ops.lrelu_mask_mul(gy, mask_ref, slope), None
    if want_x and want_w:
        acc = ops.GradSink.lookup(w)
        gx, gw = ops.conv4x4_bwd(gy, w, x, stride, pad, mask_ref, slope, acc=acc)
        return gx, (None if acc is not None else gw)
    gx = ops.conv4x4_dgrad(gy, w, x.shape[2], x.shape[3], stride, pad, mask_ref, slope) if want_x else None
    gw = _wgrad(w, gy, x, stride, pad, mask_ref, slope) if want_w else None
    return gx, gw


class _Conv(torch.autograd.Function):
    """y = conv(x, w) (linear, no bias).  Also the double-backward node of _Dgrad / _Wgrad, where `x` or `w` is a gradient
    of arbitrary magnitude: always the exact fp32-MFMA path (any_scale), never the unscaled-fp16-limb large-batch kernel."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad)
        return ops.conv4x4_fwd(x, w, None, stride, pad, 1.0, any_scale=True)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        if FUSED_BWD and not torch.is_grad_enabled():
            return _bwd_plain(ctx, w, gy, x, stride, pad) + (None, None)
        gx = _Dgrad.apply(gy, w, x.shape[2], x.shape[3], stride, pad) if ctx.needs_input_grad[0] else None
        gw = _wgrad(w, gy, x, stride, pad) if _want_w(ctx) else None
        return gx, gw, None, None


class _Dgrad(torch.autograd.Function):
    """gx = conv_transpose(g, w): bilinear in (g, w)."""

    @staticmethod
    def forward(ctx, g, w, H, W, stride, pad):
        ctx.save_for_backward(g, w)
        ctx.cfg = (stride, pad)
        return ops.conv4x4_dgrad(g, w, H, W, stride, pad)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        d_g = _Conv.apply(ggx, w, stride, pad) if ctx.needs_input_grad[0] else None
        d_w = _wgrad(w, g, ggx, stride, pad) if ctx.needs_input_grad[1] else None
        return d_g, d_w, None, None, None, None


class _Wgrad(torch.autograd.Function):
    """gw = correlate(x, g): bilinear in (g, x)."""

    @staticmethod
    def forward(ctx, g, x, stride, pad):
        ctx.save_for_backward(g, x)
        ctx.cfg = (stride, pad)
        return ops.conv4x4_wgrad(g, x, stride, pad)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        stride, pad = ctx.cfg
        d_g = _Conv.apply(x, ggw, stride, pad) if ctx.needs_input_grad[0] else None
        d_x = _Dgrad.apply(g, ggw, x.shape[2], x.shape[3], stride, pad) if ctx.needs_input_grad[1] else None
        return d_g, d_x, None, None


class _MaskMul(torch.autograd.Function):
    """out = ref > 0 ? v : slope * v; linear in v, piecewise constant in ref."""

    @staticmethod
    def forward(ctx, v, ref, slope):
        ctx.save_for_backward(ref)
        ctx.slope = slope
        return ops.lrelu_mask_mul(v, ref, slope)

    @staticmethod
    def backward(ctx, g):
        (ref,) = ctx.saved_tensors
        return _MaskMul.apply(g, ref, ctx.slope), None, None


class _ConvLrelu(torch.autograd.Function):
    """Fused forward (one kernel); backward re-expressed with the differentiable pieces above."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, slope):
        y = ops.conv4x4_fwd(x, w, None, stride, pad, slope)
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (stride, pad, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, slope = ctx.cfg
        if FUSED_BWD and not torch.is_grad_enabled():
            # plain backward: the LeakyReLU mask is applied to gy on load by both gradient kernels (no g_pre round trip)
            return _bwd_plain(ctx, w, gy, x, stride, pad, y, slope) + (None, None, None)
        g_pre = _MaskMul.apply(gy, y, slope)
        gx = _Dgrad.apply(g_pre, w, x.shape[2], x.shape[3], stride, pad) if ctx.needs_input_grad[0] else None
        gw = _Wgrad.apply(g_pre, x, stride, pad) if _want_w(ctx) else None
        return gx, gw, None, None, None


class _ConvPre(torch.autograd.Function):
    """u = conv(lrelu_{slope_in}(x), w): a layer fed with the PRE-activation of its predecessor (slope_in = 1: plain input).
    The chain form of the no-grad forward (oi_conv4x4_fwd_into) with autograd: a layer's LeakyReLU is applied by its consumer
    on load, so the split-K layers need no activation pass forward and no mask pass backward.  Plain backward: ONE launch
    (oi_conv4x4_bwd_pre: weight gradient with lrelu(x) formed on load, data gradient times lrelu'(x) in the epilogue).
    create_graph backward (the R1 inner pass): the differentiable pieces, as before."""

    @staticmethod
    def forward(ctx, x, w, slope_in, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.cfg = (slope_in, stride, pad)
        return ops.conv4x4_fwd(x, w, None, stride, pad, 1.0, x_slope=slope_in)   # (activations, not gradients: either kernel)

    @staticmethod
    def backward(ctx, gu):
        x, w = ctx.saved_tensors
        slope_in, stride, pad = ctx.cfg
        want_x, want_w = ctx.needs_input_grad[0], _want_w(ctx)
        if not torch.is_grad_enabled():
            if want_x and want_w:
                acc = ops.GradSink.lookup(w)
                gx, gw = ops.conv4x4_bwd(gu, w, x, stride, pad, acc=acc, x_slope=slope_in)
                return gx, (None if acc is not None else gw), None, None, None
            gx = ops.conv4x4_dgrad_pre(gu, w, x, slope_in, stride, pad) if want_x else None
            gw = None
            if want_w:
                a = x if slope_in == 1.0 else ops.lrelu_mask_mul(x, x, slope_in)
                gw = _wgrad(w, gu, a, stride, pad)
            return gx, gw, None, None, None
        gx = None
        if want_x:
            gx = _Dgrad.apply(gu, w, x.shape[2], x.shape[3], stride, pad)
            if slope_in != 1.0:
                gx = _MaskMul.apply(gx, x, slope_in)
        gw = None
        if want_w:
            gw = _Wgrad.apply(gu, x if slope_in == 1.0 else _MaskMul.apply(x, x, slope_in), stride, pad)
        return gx, gw, None, None, None


def conv4x4_pre(x, w, slope_in, stride, pad):
    return _ConvPre.apply(x, w, float(slope_in), stride, pad)


class _ChannelSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g):
        ctx.shape = g.shape
        return ops.channel_sum(g)

    @staticmethod
    def backward(ctx, gg):
        return gg.view(1, -1, 1, 1).expand(ctx.shape)


class _AddBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, b):
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        return g, (_ChannelSum.apply(g) if _want_w(ctx) else None)


def conv4x4_lrelu_autograd(x, w, bias, stride, pad, slope):
    if bias is None and slope != 1.0:
        return _ConvLrelu.apply(x, w, stride, pad, float(slope))
    y = _Conv.apply(x, w, stride, pad)
    if bias is not None:
        y = _AddBias.apply(y, bias)
    if slope != 1.0:
        y = _MaskMul.apply(y, y.detach(), float(slope))
    return y
