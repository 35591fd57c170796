"""Autograd Functions of the discriminator-side HIP ops.  Every backward is expressed with the same
family of kernels, itself as a Function, so gradients of arbitrary order exist -- the R1 penalty
(src/loss/gan.py:5-14) differentiates D twice.  Mirrors how the reference structures
upfirdn2d (upfirdn2d.py:214-268) and grid_sample_gradfix (grid_sample_gradfix.py:52-97)."""
import math

import torch

from . import ops


# ------------------------------------------------------------------------------------------
# upfirdn2d: linear; adjoint = upfirdn2d with up<->down swapped, flipped filter (upfirdn2d.py:243-262)
# ------------------------------------------------------------------------------------------
class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
        ctx.cfg = (upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
        ctx.x_shape = x.shape
        ctx.save_for_backward(f)
        return ops.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        upx, upy, downx, downy, px0, px1, py0, py1, flip, gain = ctx.cfg
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = dy.shape
        fh, fw = f.shape
        p = (fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1)
        dx = _Upfirdn2d.apply(dy, f, downx, downy, upx, upy, p[0], p[1], p[2], p[3], not flip, gain)
        return (dx,) + (None,) * 11


def upfirdn2d(x, f2d, upx=1, upy=1, downx=1, downy=1, px0=0, px1=0, py0=0, py1=0, flip=False, gain=1.0):
    return _Upfirdn2d.apply(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, bool(flip), float(gain))


def upfirdn2d_separable(x, f1, up=1, down=1, pad=(0, 0, 0, 0), flip=False, gain=1.0):
    """Two 1-D passes with sqrt(gain) each, as the reference plugin path (upfirdn2d.py:239-241)."""
    g = math.sqrt(gain)
    y = upfirdn2d(x, f1[None, :], up, 1, down, 1, pad[0], pad[1], 0, 0, flip, g)
    return upfirdn2d(y, f1[:, None], 1, up, 1, down, 0, 0, pad[2], pad[3], flip, g)


# ------------------------------------------------------------------------------------------
# affine grid sample (theta carries no gradient on the path: augmentation parameters are sampled)
# ------------------------------------------------------------------------------------------
class _AffineGridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, theta, Ho, Wo):
        ctx.save_for_backward(theta)
        ctx.in_hw = x.shape[2:]
        return ops.affine_grid_sample_fwd(x, theta, Ho, Wo)

    @staticmethod
    def backward(ctx, gy):
        (theta,) = ctx.saved_tensors
        return _AffineGridSampleBwd.apply(gy, theta, ctx.in_hw[0], ctx.in_hw[1]), None, None, None


class _AffineGridSampleBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, theta, Hi, Wi):
        ctx.save_for_backward(theta)
        ctx.out_hw = gy.shape[2:]
        return ops.affine_grid_sample_bwd(gy, theta, Hi, Wi)

    @staticmethod
    def backward(ctx, ggx):
        (theta,) = ctx.saved_tensors
        return _AffineGridSample.apply(ggx, theta, ctx.out_hw[0], ctx.out_hw[1]), None, None, None


def affine_grid_sample(x, theta, Ho, Wo):
    return _AffineGridSample.apply(x, theta.detach(), Ho, Wo)


# ------------------------------------------------------------------------------------------
# reflect pad
# ------------------------------------------------------------------------------------------
class _ReflectPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, px0, px1, py0, py1):
        ctx.cfg = (x.shape[2], x.shape[3], px0, px1, py0, py1)
        return ops.reflect_pad_fwd(x, px0, px1, py0, py1)

    @staticmethod
    def backward(ctx, gy):
        return (_ReflectPadBwd.apply(gy, *ctx.cfg),) + (None,) * 4


class _ReflectPadBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, H, W, px0, px1, py0, py1):
        ctx.cfg = (px0, px1, py0, py1)
        return ops.reflect_pad_bwd(gy, H, W, px0, px1, py0, py1)

    @staticmethod
    def backward(ctx, ggx):
        return (_ReflectPad.apply(ggx, *ctx.cfg),) + (None,) * 6


def reflect_pad(x, px0, px1, py0, py1):
    if px0 == px1 == py0 == py1 == 0:
        return x
    return _ReflectPad.apply(x, px0, px1, py0, py1)


# ------------------------------------------------------------------------------------------
# the whole geometric augmentation (reflect pad -> x2 up-FIR -> affine resample -> /2 down-FIR) as ONE node: two launches
# forward (oi_ada_geom_fwd).  The map is linear in the images, so nothing is saved but theta; the backward is the chain of
# the four adjoint Functions above (themselves differentiable: the R1 double backward comes back through the forward ops).
# ------------------------------------------------------------------------------------------
def _ada_geom_adjoint(gy, theta, f1, H, W, margins):
    """A^T gy with the raw kernels (no autograd nodes: the caller is a Function whose own backward is the forward map)."""
    mx0, my0, mx1, my1 = margins
    Hp, Wp = H + my0 + my1, W + mx0 + mx1
    Hz_pad = f1.shape[0] // 4
    Ho, Wo = (H + Hz_pad * 2) * 2, (W + Hz_pad * 2) * 2
    n = f1.shape[0]
    fx, fy = f1[None, :], f1[:, None]
    # adjoint of upfirdn2d(x, f, up, down, pads, flip, g) on an input of `size`: upfirdn2d with up <-> down, the other flip
    # and the pads of upfirdn2d.py:243-262 -- one axis at a time, in reverse order of the forward's x-then-y passes
    def adj_y(g, up, down, p0, size_in, flip, gain):
        return ops.upfirdn2d(g, fy, 1, down, 1, up, 0, 0, n - p0 - 1, size_in * up - g.shape[2] * down + p0 - up + 1,
                             not flip, gain)

    def adj_x(g, up, down, p0, size_in, flip, gain):
        return ops.upfirdn2d(g, fx, down, 1, up, 1, n - p0 - 1, size_in * up - g.shape[3] * down + p0 - up + 1, 0, 0,
                             not flip, gain)

    g = adj_y(gy, 1, 2, -1, Ho, True, 1.0)          # downsample2d(padding = -2 Hz_pad, flip_filter): pads (-1, -1)
    g = adj_x(g, 1, 2, -1, Wo, True, 1.0)
    g = ops.affine_grid_sample_bwd(g, theta, 2 * Hp, 2 * Wp)
    g = adj_y(g, 2, 1, 6, Hp, False, 2.0)           # upsample2d: pads (6, 5), gain 4 = 2 per axis
    g = adj_x(g, 2, 1, 6, Wp, False, 2.0)
    return ops.reflect_pad_bwd(g, H, W, mx0, mx1, my0, my1)


class _AdaGeom(torch.autograd.Function):
    """y = A(theta) x.  backward: A^T (as _AdaGeomAdjoint); the backward of THAT is this forward again -- so the double
    backward an R1 penalty takes runs the two fused launches too, not six separate stages."""

    @staticmethod
    def forward(ctx, x, theta, f1, margins, axis_aligned=False):
        ctx.save_for_backward(theta, f1)
        ctx.margins, ctx.axis_aligned = tuple(margins), bool(axis_aligned)
        return ops.ada_geom_fwd(x, theta, f1, margins, axis_aligned)

    @staticmethod
    def backward(ctx, gy):
        theta, f1 = ctx.saved_tensors
        return _AdaGeomAdjoint.apply(gy, theta, f1, ctx.margins, ctx.axis_aligned), None, None, None, None


class _AdaGeomAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, theta, f1, margins, axis_aligned=False):
        ctx.save_for_backward(theta, f1)
        ctx.margins, ctx.axis_aligned = tuple(margins), bool(axis_aligned)
        if axis_aligned and ops.ada_geom_sep_ok(gy):
            return ops.ada_geom_adj_sep(gy, theta, f1, margins)   # one launch: A_y^T gy A_x
        return _ada_geom_adjoint(gy, theta, f1, gy.shape[2], gy.shape[3], margins)

    @staticmethod
    def backward(ctx, ggx):
        theta, f1 = ctx.saved_tensors
        return _AdaGeom.apply(ggx, theta, f1, ctx.margins, ctx.axis_aligned), None, None, None, None


def ada_geom(x, theta, f1, margins, axis_aligned=False):
    """`axis_aligned`: the caller's promise that no theta carries a rotation (the one-launch separable form, ops.ada_geom_fwd)."""
    return _AdaGeom.apply(x, theta.detach(), f1, tuple(int(m) for m in margins), bool(axis_aligned))


# ------------------------------------------------------------------------------------------
# conv 4x4 (+ fused LeakyReLU)
# ------------------------------------------------------------------------------------------
def conv4x4_lrelu(x, w, bias, stride, pad, slope):
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad)):
        from .autograd_conv import conv4x4_lrelu_autograd
        return conv4x4_lrelu_autograd(x, w, bias, stride, pad, slope)
    return ops.conv4x4_fwd(x, w, bias, stride, pad, slope)
