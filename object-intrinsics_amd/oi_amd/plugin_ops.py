"""The reference's compiled plugin ops under their own names and argument lists, backed by liboi_hip.so -- for a
maintainer who keeps the reference's Python layers (stylesdf/op/fused_act.py, ada/torch_utils/ops/*.py) and only swaps
what those import.

    fused_bias_act(input, bias, refer, act, grad, alpha, scale)   stylesdf/op/fused_bias_act.cpp:11-20
    upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)   stylesdf/op/upfirdn2d.cpp:14-26
    grid_sample(input, grid)                                       ada/torch_utils/ops/grid_sample_gradfix.py:33-66

fp32 only (the path never runs these in half precision: train.yaml has no fp16 switch).  No CPU fallback: a missing
library raises (oi_amd.lib)."""
import torch

from . import lib as _l
from . import ops as _ops

_p = _ops._p
_c = _ops._c


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Same contract as the reference op: `bias` / `refer` may be empty tensors; the bias runs over dimension 1."""
    x = _c(input)
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    b = _c(bias) if bias is not None and bias.numel() else None
    r = _c(refer) if refer is not None and refer.numel() else None
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    _l.check(_l.load().oi_fused_bias_act(_p(out), _p(x), _p(b), _p(r), int(act), int(grad), float(alpha), float(scale),
                                         x.numel(), step_b, b.numel() if b is not None else 1, _ops._stream()),
             "oi_fused_bias_act")
    return out


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """The plugin's argument list; NCHW in, NCHW out (upfirdn2d.py:160-186 reshapes around the same call)."""
    return _ops.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)


def _gs_fwd(x, grid):
    N, C, Hi, Wi = x.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    y = torch.empty(N, C, Ho, Wo, dtype=x.dtype, device=x.device)
    _l.check(_l.load().oi_grid_sample_fwd(_p(_c(x)), _p(_c(grid)), _p(y), N, C, Hi, Wi, Ho, Wo, _ops._stream()),
             "oi_grid_sample_fwd")
    return y


def _gs_bwd(gy, x, grid, need_x, need_grid):
    N, C, Hi, Wi = x.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    gx = _ops._new_acc(x, *x.shape) if need_x else None   # a scatter-add target: pre-zeroed when a ZeroPool is active
    gg = torch.empty_like(grid) if need_grid else None
    _l.check(_l.load().oi_grid_sample_bwd(_p(_c(gy)), _p(_c(x)), _p(_c(grid)), _p(gx), _p(gg), N, C, Hi, Wi, Ho, Wo,
                                          _ops._stream()), "oi_grid_sample_bwd")
    return gx, gg


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4 and grid.shape[-1] == 2
        ctx.save_for_backward(input, grid)
        return _gs_fwd(input, grid)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return _GridSample2dBackward.apply(grad_output, input, grid)


class _GridSample2dBackward(torch.autograd.Function):
    """First-order backward as its own Function so that R1-style double backward works: d grad_input / d grad_output is
    grid_sample itself (the same structure as grid_sample_gradfix.py:55-83; no second derivative w.r.t. the grid)."""

    @staticmethod
    def forward(ctx, grad_output, input, grid):
        ctx.save_for_backward(grid)
        gx, gg = _gs_bwd(grad_output, input, grid, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gx, gg

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grid, = ctx.saved_tensors
        g2 = None
        if ctx.needs_input_grad[0] and grad2_grad_input is not None:
            g2 = _GridSample2dForward.apply(grad2_grad_input, grid)
        return g2, None, None


def grid_sample(input, grid):
    """bilinear, zeros padding, align_corners=False (the only mode the reference uses)."""
    return _GridSample2dForward.apply(input, grid)
