"""DC discriminators -- drop-ins for src.models.discriminator.DCDiscriminator / ADADiscriminator /
ADADiscriminatorView (discriminator.py:49-108): same constructor kwargs, `forward(x, **kwargs)`,
`get_resolution()`, `.aug`, and state_dict keys `blocks.{i}.weight`, `conv_out.weight[/bias]`,
`aug.p`, `aug.Hz_geom`, `aug.Hz_fbank`.  Convolutions are the implicit-GEMM MFMA kernels of
csrc/disc.hip with the LeakyReLU fused."""
from math import log2

import torch
import torch.nn as nn

from .autograd_disc import conv4x4_lrelu
from .config import build_from_config

import weakref

_SMALL_PLANS = weakref.WeakKeyDictionary()
_LARGE_PACKS = weakref.WeakKeyDictionary()
LARGE_PATH = True   # batch >= 16 no-grad forwards take csrc/disc_large.hip when the network is covered (False: the general chain)
LARGE_MIN_BATCH = 16
FAST_ADA = True     # ADADiscriminator.forward, shipped augmentation, batch <= 4: parameters drawn inside the library (False: numpy)
SMALL_PATH = True   # batch <= 4 no-grad forwards of the 64 x 64 network take csrc/disc_small.hip (False: the general chain)
import os
SMALL_PATH_128 = os.environ.get("OI_SMALL128", "1") != "0"   # ... and of the shipped 128 x 128 / five-block network (round 6)


class _ConvParam(nn.Module):
    """Holds `weight` (and optional `bias`) under the same names as nn.Conv2d(…, 4, s, p)."""

    def __init__(self, cin, cout, bias=False):
        super().__init__()
        conv = nn.Conv2d(cin, cout, 4, bias=bias)  # identical default initialisation to the reference's layers
        self.weight = nn.Parameter(conv.weight.detach().clone())
        if bias:
            self.bias = nn.Parameter(conv.bias.detach().clone())
        else:
            self.register_parameter("bias", None)


class DCDiscriminator(nn.Module):
    def __init__(self, in_dim=3, out_dim=1, n_feat=512, img_size=64, last_bias=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        n_layers = int(log2(img_size) - 2)
        chans = [in_dim] + [int(n_feat / (2 ** (n_layers - 1 - i))) for i in range(n_layers)]
        self.blocks = nn.ModuleList([_ConvParam(chans[i], chans[i + 1]) for i in range(n_layers)])
        self.conv_out = _ConvParam(n_feat, out_dim, bias=last_bias)

    def _forward_nograd(self, x):
        """Forward-only chain: all layer outputs live in ONE zero-filled arena (a single fill launch instead of one per
        split-K layer) and every block hands over its pre-activation sums -- the LeakyReLU is applied by the next
        layer while it loads them, so the split-K layers need no activation pass: 6 launches instead of 13."""
        from . import ops
        # (never under somebody's stream capture: the packed weights and the workspace are eager allocations keyed on the
        #  parameter versions -- a replay after an optimiser step would read the stale pack, and the next eager call frees it
        #  under the graph (advisor, round 5).  The general chain below reads the live parameters.)
        if LARGE_PATH and x.shape[0] >= LARGE_MIN_BATCH and x.is_cuda and not torch.cuda.is_current_stream_capturing():
            out = self._forward_large(x)
            if out is not None:
                return out
        layers = [(l.weight, None, 2, 1, 0.2) for l in self.blocks] + [(self.conv_out.weight, self.conv_out.bias, 1, 0, 1.0)]
        shapes, shp = [], tuple(x.shape)
        for w, _, stride, pad, _ in layers:
            shp = ops.conv4x4_out_shape(shp, w.shape[0], stride, pad)
            shapes.append(shp)
        sizes = [(s[0] * s[1] * s[2] * s[3] + 3) // 4 * 4 for s in shapes]
        # no fill launch: the first layer (K = 16 in_dim taps: never split) clears the rest of the arena while it runs
        arena = torch.empty(sum(sizes), dtype=torch.float32, device=x.device)
        off = 0
        x_slope = 1.0
        for i, ((w, b, stride, pad, slope), s, n) in enumerate(zip(layers, shapes, sizes)):
            last = i == len(layers) - 1
            x = ops.conv4x4_fwd(x, w, b, stride, pad, slope if last else 1.0, x_slope=x_slope,
                                out=arena[off:off + s[0] * s[1] * s[2] * s[3]].view(s),
                                zero_tail=sum(sizes) - sizes[0] if i == 0 else 0, out_is_zero=i > 0)
            x_slope = slope  # this block's activation, deferred to the next layer's loads
            off += n
        return x

    def _forward_large(self, x):
        """Batch >= 16: activations as NHWC fp16 limb planes, weights packed once per parameter version (ops.DiscLargePack), K split
        with a fixed-order reduction -- csrc/disc_large.hip.  None when the network / shape is not covered."""
        from . import ops
        ws = [l.weight for l in self.blocks] + [self.conv_out.weight]
        key = tuple((w.data_ptr(), w._version) for w in ws)
        ent = _LARGE_PACKS.get(self)
        if ent is None or ent[0] != key:
            ent = _LARGE_PACKS[self] = (key, ops.DiscLargePack(ws[:-1], ws[-1]))
        return ops.disc_fwd_large(x, ent[1], ws[0], self.conv_out.bias)

    def _small_ok(self, x):
        """The four- / five-launch forward of csrc/disc_small.hip: no gradient, batch <= 4, the 64 x 64 / n_feat 512 network, fp32
        contiguous weights with the input's channel count (the kernels index them by shape), Callers that would
        create a library plan also ask `_plan_ok()`: a plan's arrival counters are zeroed when the plan is created, and a fill
        recorded into someone else's stream capture never runs for the eager calls that follow (advisor, round 4) -- under capture
        the general chain is taken instead; the plain entry with a device-resident matrix (`theta_dev`, what captured graphs use)
        refuses a FIRST call inside a capture (ops.disc_fwd_small)."""
        if not (SMALL_PATH and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.shape[0] <= 4 and x.shape[1] <= 4
                and self.out_dim <= 8):
            return False
        # BASELINE's 64 x 64 network (four blocks 64 .. 512) or the shipped 128 x 128 one (five blocks 32 .. 512: configs/train.yaml:78-102)
        hw, nb = tuple(x.shape[2:]), len(self.blocks)
        if not ((hw == (64, 64) and nb == 4) or (hw == (128, 128) and nb == 5 and SMALL_PATH_128)):
            return False
        if [int(l.weight.shape[0]) for l in self.blocks] != ([64, 128, 256, 512] if nb == 4 else [32, 64, 128, 256, 512]):
            return False
        ws = [l.weight for l in self.blocks] + [self.conv_out.weight] + ([] if self.conv_out.bias is None else [self.conv_out.bias])
        if not (x.shape[1] == self.in_dim == self.blocks[0].weight.shape[1]
                and all(w.dtype == torch.float32 and w.is_contiguous() for w in ws)):
            return False
        return True

    @staticmethod
    def _plan_ok():
        """Library plans (ops.DiscGraph) are created lazily: never inside somebody's stream capture (see _small_ok)."""
        return not torch.cuda.is_current_stream_capturing()

    def _forward_small(self, x, f12=None, theta_np=None, theta_dev=None, margins=None):
        """theta_dev (captured graphs): the plain entry.  Otherwise a plan held by the library (ops.DiscGraph, launch by launch):
        everything but the image pointer and the matrices is marshalled once per (shape, stream, weight addresses), not per
        call -- 50 us of host time per forward less."""
        from . import ops
        if theta_dev is not None:
            return ops.disc_fwd_small(x.float(), [l.weight for l in self.blocks], self.conv_out.weight, self.conv_out.bias, f12=f12,
                                      theta_dev=theta_dev, margins=margins)
        aug_on = theta_np is not None
        return self._small_plan(x, f12 if aug_on else None, margins if aug_on else None)(x.float(), theta_np, fresh=True)

    def _small_plan(self, x, f12, margins):
        """The library plan (ops.DiscGraph, launched launch by launch) for this image shape / stream / weight addresses; with the
        augmentation at static `margins` when they are given."""
        from . import ops
        ws = [l.weight for l in self.blocks] + [self.conv_out.weight] + ([] if self.conv_out.bias is None else [self.conv_out.bias])
        key = (tuple(x.shape), x.device.index, ops._stream().value or 0, None if margins is None else tuple(int(v) for v in margins),
               tuple(w.data_ptr() for w in ws))
        plans = _SMALL_PLANS.setdefault(self, {})   # (not in __dict__: the plans hold library handles; copy.deepcopy(module) must work)
        plan = plans.get(key)
        if plan is None:
            if len(plans) >= 8:
                plans.clear()   # (weights moved, many shapes: start over rather than grow)
                _FAST_ADA.pop(self, None)
            plan = plans[key] = ops.DiscGraph(tuple(x.shape), x.device, [l.weight for l in self.blocks], self.conv_out.weight,
                                              self.conv_out.bias, f12=f12, margins=margins, launch="eager")
        return plan

    def forward(self, x, **kwargs):
        batch_size = x.shape[0]
        assert x.shape[1] == self.in_dim, x.shape
        if self._small_ok(x) and self._plan_ok():
            return self._forward_small(x)
        if not torch.is_grad_enabled():
            return self._forward_nograd(x.float()).reshape(batch_size, self.out_dim)
        from . import autograd_conv as AC
        if AC.PRE_CHAIN and x.is_cuda:
            # pre-activation chain (autograd_conv._ConvPre): every block hands over its sums, the next layer applies the
            # LeakyReLU while it loads them -- no activation pass forward, no mask pass backward
            slope_in = 1.0
            for layer in self.blocks:
                x = AC.conv4x4_pre(x, layer.weight, slope_in, 2, 1)
                slope_in = 0.2
            out = AC.conv4x4_pre(x, self.conv_out.weight, slope_in, 1, 0)
            if self.conv_out.bias is not None:
                out = AC._AddBias.apply(out, self.conv_out.bias)
            return out.reshape(batch_size, self.out_dim)
        for layer in self.blocks:
            x = conv4x4_lrelu(x, layer.weight, None, stride=2, pad=1, slope=0.2)
        out = conv4x4_lrelu(x, self.conv_out.weight, self.conv_out.bias, stride=1, pad=0, slope=1.0)
        return out.reshape(batch_size, self.out_dim)


class _FastSmallAda:
    """What ADADiscriminator.forward needs to answer a no-grad batch <= 4 call of the shipped configuration with ONE library call
    (ops.DiscGraph.call_ada: draws, sampling matrices, the four launches): the plan and the guards under which it stays valid --
    shape / dtype / layout of the image, the stream, the addresses of the weights (the plan holds raw pointers; in-place
    optimiser updates keep them), nobody capturing, the augmentation pipe un-patched.  Any guard failing -> None -> the general
    path decides again (and may build a new one).  Review, round 5: the module's own forward ran at 38 % of the plan's rate --
    numpy draws (20 us), matrix algebra (4 us), array marshalling (5 us) and ~15 us of per-call checks for 16 us of launches."""

    __slots__ = ("shape", "device", "stream", "ws", "ptrs", "plan", "aug")

    def __init__(self, disc, x, plan):
        from . import ops
        self.shape, self.device, self.stream = x.shape, x.device, ops._stream().value
        self.ws = [l.weight for l in disc.blocks] + [disc.conv_out.weight] + ([] if disc.conv_out.bias is None else [disc.conv_out.bias])
        self.ptrs = [w.data_ptr() for w in self.ws]
        self.plan, self.aug = plan, disc.aug

    def __call__(self, x):
        if (x.shape != self.shape or x.dtype is not torch.float32 or x.device != self.device or not x.is_contiguous()
                or [w.data_ptr() for w in self.ws] != self.ptrs):
            return None
        d = self.aug.__dict__
        if "sample_G_inv" in d or "forward" in d or torch.cuda.is_current_stream_capturing():
            return None
        from . import ops
        if ops._stream().value != self.stream:
            return None
        return self.plan.call_ada(x, self.aug.draw_seed(), *self.aug.fast_params(), fresh=True)


_FAST_ADA = weakref.WeakKeyDictionary()


class ADADiscriminator(DCDiscriminator):
    def __init__(self, aug, aug_p, **kwargs):
        super().__init__(**kwargs)
        self.aug = build_from_config(aug)
        self.aug.p.copy_(torch.tensor(aug_p, dtype=torch.float32))
        self.resolution = kwargs["img_size"]

    def get_resolution(self):
        return self.resolution

    def forward(self, x, aug_theta=None, **kwargs):
        """`aug_theta`: precomputed sampling grid for the shape-static augmentation (see AugmentPipe.forward)."""
        from .augment import AugmentPipe
        aug = self.aug
        if aug_theta is None and not torch.is_grad_enabled():
            fast = _FAST_ADA.get(self)
            if fast is not None:
                out = fast(x)
                if out is not None:
                    return out
        if (self._small_ok(x) and type(aug).forward is AugmentPipe.forward and "forward" not in aug.__dict__
                and aug.Hz_geom.shape[0] == 12):
            # augmentation + network in four launches; the sampling matrix goes to the kernel by value (no upload)
            H, W = x.shape[2:]
            if aug_theta is not None:
                return self._forward_small(x, f12=aug.Hz_geom, theta_dev=aug_theta, margins=aug.static_margins(H, W))
            if not self._plan_ok():
                return super().forward(self.aug(x), **kwargs)
            if FAST_ADA and aug.fast_draw_ok():
                # the shipped configuration: parameters drawn inside the library from one seed of numpy's stream
                x = x.float().contiguous()
                margins = aug.static_margins(H, W)
                plan = self._small_plan(x, aug.Hz_geom, margins)
                fast = _FAST_ADA[self] = _FastSmallAda(self, x, plan)
                return fast(x)
            G_inv = aug.sample_G_inv(x, None)
            if G_inv is None:
                return self._forward_small(x)
            # the largest margins the transform's own clamp allows (static: one plan serves every draw; the same pixels are
            # sampled as with margins fitted to the draw, augment.py:272-282)
            margins = aug.static_margins(H, W)
            return self._forward_small(x, f12=aug.Hz_geom, theta_np=aug.theta_for(G_inv, margins, H, W), margins=margins)
        if (aug_theta is None and FAST_ADA and not torch.is_grad_enabled() and x.is_cuda and x.shape[0] >= LARGE_MIN_BATCH
                and type(aug).forward is AugmentPipe.forward and aug.fast_draw_ok()):
            # large no-grad batches of the shipped configuration: the parameters from one seed of numpy's stream, expanded inside
            # the library (as the batch <= 4 path above), the sampling matrices in the arguments of ONE augmentation launch
            # (numpy draws + matrix algebra + the upload of 64 matrices cost the host more than the GPU needs for the forward)
            from . import ops
            xf = x.float().contiguous()
            if ops.ada_geom_sep_ok(xf):
                B, _, H, W = xf.shape
                return super().forward(ops.ada_geom_sep_host(xf, aug.theta_fast(B, H, W), aug.Hz_geom, aug.static_margins(H, W)), **kwargs)
        return super().forward(self.aug(x) if aug_theta is None else self.aug(x, theta=aug_theta), **kwargs)


class ADADiscriminatorView(ADADiscriminator):
    def __init__(self, out_dim_position, out_dim_latent, **kwargs):
        self.out_dim_position = out_dim_position
        self.out_dim_latent = out_dim_latent
        super().__init__(**kwargs)
