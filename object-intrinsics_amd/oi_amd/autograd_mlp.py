"""autograd.Function of the FiLM-SIREN MLP (forward csrc/mlp.hip, backward csrc/mlp_bwd.hip).

The Function takes the *stacked* parameter views (oi_amd.params.stack_field_params) as explicit
inputs, so the gradients the HIP backward produces flow on through torch.stack to the
reference-named nn.Parameters.  The backward already contains the second-order terms of the
forward's d sdf/dx output, so the Function itself is once-differentiable."""
import torch
from torch.autograd.function import once_differentiable

from . import ops

_PKEYS = ("w0", "b0", "wh", "bh", "wsig", "bsig", "wv", "bv", "wrgb", "brgb")


class SdfMlpFunction(torch.autograd.Function):
    @staticmethod
    def run(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat):
        P = pack.stacked()
        out = SdfMlpFunction.apply(pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat, *[P[k] for k in _PKEYS])
        return out

    @staticmethod
    def forward(ctx, pack, pts, gamma, beta, B, want_grad, want_rgb, want_feat, *params):
        packed = pack.packed()
        # the backward differentiates the albedo head first, from the features (a_8) the forward wrote
        sdf, grad, rgb, feat, _ = ops.sdf_mlp_fwd(pts, packed, gamma, beta, B, pack.prec, pack.fast_trig, want_grad,
                                                  want_rgb, want_feat or want_rgb)
        ctx.pack, ctx.B = pack, B
        ctx.set_materialize_grads(False)   # an output no loss touches: None -> null pointer, not a zero-filled tensor
        # the image of the weights the forward used (parameters may be stepped before backward is called)
        ctx.packed = packed if pack.prec_bwd == pack.prec else pack.packed(for_backward=True)
        ctx.save_for_backward(pts, gamma, beta, grad, rgb, feat)
        if feat is not None and not want_feat:
            ctx.mark_non_differentiable(feat)
        # a feature output the caller asked for is differentiable: its gradient (ColorNetwork.forward on the features read out,
        # the reference's renderer.py:241-261) joins abar_8 in the sweep (oi_sdf_mlp_bwd_feat)
        return sdf, grad, rgb, (feat if want_feat else None)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_sdf, g_grad, g_rgb, g_feat):
        pts, gamma, beta, grad, rgb, feat = ctx.saved_tensors
        pack = ctx.pack
        d_small, d_wmat, d_gamma, d_beta = ops.sdf_mlp_bwd(pts, ctx.packed, gamma, beta, grad, rgb, feat, g_sdf, g_grad,
                                                           g_rgb if rgb is not None else None, ctx.B, pack.prec_bwd,
                                                           pack.fast_trig, g_feat=g_feat)
        s = d_small
        d_w0 = s[0:384].view(128, 3)
        d_b = s[384:1536].view(9, 128)
        d_wsig, d_bsig = s[1536:1664], s[1664:1665]
        d_wvx = s[1668:2052].view(128, 3)
        d_wrgb, d_brgb = s[2052:2436].view(3, 128), s[2436:2439]
        d_wv = torch.cat([d_wmat[7], d_wvx], dim=1)
        grads = (d_w0, d_b[0], d_wmat[:7], d_b[1:8], d_wsig, d_bsig, d_wv, d_b[8], d_wrgb, d_brgb)
        return (None, None, d_gamma, d_beta, None, None, None, None) + grads
