"""Latent-conditioned FiLM-SIREN SDF / albedo networks -- drop-ins for the reference classes
    src.models.fields.ShapeNetwork / ColorNetwork            (src/models/fields.py:10-101)
    src.third_party.neus.models.fields.SingleVarianceNetwork (neus/models/fields.py:262-268)
with identical constructor kwargs, method names and state_dict keys:
    style.{0,1,2}.{weight,bias}; pts_linears.{l}.{weight,bias,gamma.weight,gamma.bias,beta.weight,beta.bias};
    sigma_linear.{weight,bias}; views_linears.{...}; rgb_linear.{weight,bias}; variance.
The modules only *hold* parameters; evaluation happens in the HIP MLP kernel (csrc/mlp.hip) through
`FieldPack`, which caches the MFMA weight images per parameter version."""

import numpy as np
import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .params import stack_field_params


class LinearLayer(nn.Module):
    """std_init * (x W^T + b) + bias_init   (stylesdf/volume_renderer.py:12-30)."""

    def __init__(self, in_dim, out_dim, bias_init=0, std_init=1, freq_init=False, is_first=False):
        super().__init__()
        if is_first:
            w = torch.empty(out_dim, in_dim).uniform_(-1 / in_dim, 1 / in_dim)
        elif freq_init:
            a = np.sqrt(6 / in_dim) / 25
            w = torch.empty(out_dim, in_dim).uniform_(-a, a)
        else:
            w = 0.25 * nn.init.kaiming_normal_(torch.randn(out_dim, in_dim), a=0.2, mode="fan_in",
                                               nonlinearity="leaky_relu")
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_dim).uniform_(-np.sqrt(1 / in_dim), np.sqrt(1 / in_dim)))
        self.bias_init, self.std_init = bias_init, std_init


class FiLMSiren(nn.Module):
    """sin(gamma(w) * (x W^T + b) + beta(w))   (stylesdf/volume_renderer.py:33-61)."""

    def __init__(self, in_channel, out_channel, style_dim, is_first=False):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        a = 1 / 3 if is_first else np.sqrt(6 / in_channel) / 25
        self.weight = nn.Parameter(torch.empty(out_channel, in_channel).uniform_(-a, a))
        self.bias = nn.Parameter(torch.empty(out_channel).uniform_(-np.sqrt(1 / in_channel), np.sqrt(1 / in_channel)))
        self.gamma = LinearLayer(style_dim, out_channel, bias_init=30, std_init=15)
        self.beta = LinearLayer(style_dim, out_channel, bias_init=0, std_init=0.25)


class MappingLinear(nn.Module):
    """linear + fused leaky-relu(0.2), scale 1   (stylesdf/model.py:32-56)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.weight = nn.Parameter(nn.init.kaiming_normal_(torch.empty(out_dim, in_dim), a=0.2, mode="fan_in",
                                                           nonlinearity="leaky_relu"))
        self.bias = nn.Parameter(torch.empty(out_dim).uniform_(-np.sqrt(1 / in_dim), np.sqrt(1 / in_dim)))


class StyleMLP(nn.Sequential):
    """`ShapeNetwork.style`: three MappingLinear layers evaluated by one HIP launch (oi_film_params)."""

    def forward(self, z):
        from .autograd import style_mlp
        return style_mlp(self, z)


def _check_dims(D, W, input_ch, style_dim):
    if (D, W, input_ch, style_dim) != (8, 128, 3, 64):
        raise NotImplementedError(
            f"the gfx950 MLP kernel is specialised for D=8, W=128, input_ch=3, style_dim=64 "
            f"(configs/train.yaml:34-49); got D={D} W={W} input_ch={input_ch} style_dim={style_dim}")


class ShapeNetwork(nn.Module):
    def __init__(self, checkpoint_path, D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64, **_unused):
        super().__init__()
        _check_dims(D, W, input_ch, style_dim)
        self.style = StyleMLP(*[MappingLinear(style_dim, style_dim) for _ in range(3)])
        self.pts_linears = nn.ModuleList([FiLMSiren(input_ch, W, style_dim, is_first=True)] +
                                         [FiLMSiren(W, W, style_dim) for _ in range(D - 1)])
        self.sigma_linear = LinearLayer(W, 1, freq_init=True)
        self._pack = None
        if checkpoint_path is not None:
            self.load_state_dict(load_sdf_checkpoint(checkpoint_path))

    # -- reference API (fields.py:49-77); the renderer does not go through these -----------------
    def _own_pack(self):
        if self._pack is None:
            object.__setattr__(self, "_pack", FieldPack(self, None))
        return self._pack

    def forward(self, x, z, w=None):
        from .autograd import sdf_mlp
        pk = self._own_pack()
        w, gamma, beta = pk.film(z=z if w is None else None, w=w)
        sdf, _, _, feat = sdf_mlp(pk, x, gamma, beta, w.shape[0], want_grad=False, want_rgb=False, want_feat=True)
        return torch.cat([sdf[:, None], feat], -1)

    def sdf(self, x, z, w=None):
        from .autograd import sdf_mlp
        pk = self._own_pack()
        w, gamma, beta = pk.film(z=z if w is None else None, w=w)
        return sdf_mlp(pk, x, gamma, beta, w.shape[0], want_grad=False, want_rgb=False, want_feat=False)[0][:, None]

    def gradient(self, x, z, w=None, second_order=False):
        from .autograd import sdf_mlp
        if second_order:
            raise NotImplementedError("second_order=True (hessian) is dead on the path (renderer.py:254)")
        pk = self._own_pack()
        w, gamma, beta = pk.film(z=z if w is None else None, w=w)
        return sdf_mlp(pk, x, gamma, beta, w.shape[0], want_grad=True, want_rgb=False, want_feat=False)[1]

    def __deepcopy__(self, memo):  # EMA copies (src/utils/ema.py:11-12): never share the cache
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            object.__setattr__(new, k, None if k == "_pack" else copy.deepcopy(v, memo))
        return new


class ColorNetwork(nn.Module):
    def __init__(self, D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64, **_unused):
        super().__init__()
        _check_dims(D, W, input_ch, style_dim)
        if input_ch_views != 3:
            raise NotImplementedError("colour head consumes [feat(128), grad(3)] (fields.py:96)")
        self.views_linears = FiLMSiren(input_ch_views + W, W, style_dim)
        self.rgb_linear = LinearLayer(W, 3, freq_init=True)
        self.style_dim, self.w_dim = style_dim, W

    def forward(self, points, normals, view_dirs, feature_vectors, z=None, w=None):
        """The reference's stand-alone evaluation (fields.py:89-101): rgb = sigmoid(rgb_linear(FiLM-sin(cat[feature_vectors,
        normals]; w))) on CALLER-SUPPLIED features and normals; `points`, `view_dirs` and `z` are ignored there and here (only
        points.shape[0] is used).  One HIP launch (csrc/color_head.hip, exact fp32 on the matrix cores) behind an autograd
        Function whose backward reaches the features, the normals, w and every parameter of this module.  NeuSRenderer.render
        does NOT come through here: in the render the head is the tail of the fused MLP launch and its inputs never leave the
        kernel -- this entry is for a caller that keeps the reference's renderer.py:241-261."""
        from .autograd import color_head, FilmParamsFunction, _needs_grad
        if w is None:
            raise ValueError("ColorNetwork.forward needs the style vector w (fields.py:91: bs = w.shape[0])")
        n, B = feature_vectors.shape[0], w.shape[0]
        if points.shape[0] != n or normals.shape[0] != n or n % B:
            raise ValueError(f"ColorNetwork.forward: {n} feature rows, {points.shape[0]} points, {normals.shape[0]} normals, batch {B}")
        vl = self.views_linears
        heads = (vl.gamma.weight[None], vl.gamma.bias[None], vl.beta.weight[None], vl.beta.bias[None])   # NL = 1
        if _needs_grad(w, *heads):
            _, gamma, beta = FilmParamsFunction.apply(None, w, None, None, *heads)
        else:
            with torch.no_grad():
                _, gamma, beta = ops.film_params(None, None, *[ops._c(t) for t in heads], w=w)
        return color_head(feature_vectors.reshape(n, 128), normals.reshape(n, 3), gamma[:, 0], beta[:, 0], vl.weight, vl.bias,
                          self.rgb_linear.weight, self.rgb_linear.bias, B)


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(float(init_val)))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)


def load_sdf_checkpoint(path):
    """Reference checkpoints are torch pickles {'sdf_network': state_dict, 'cfg': ...}
    (fields.py:25-38); the test fixture tests/golden/weights_sdf.npz holds the same tensors."""
    if str(path).endswith(".npz"):
        with np.load(path) as f:
            return {k: torch.from_numpy(np.asarray(f[k])) for k in f.files}
    sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd["sdf_network"] if "sdf_network" in sd else sd


class FieldPack:
    """Per (sdf_network, color_network) cache of what the kernels consume: stacked parameter views
    (differentiable torch.stack) and the packed MFMA weight image, keyed by parameter versions."""

    def __init__(self, sdf_network, color_network, precision="f16x3", fast_trig=None):
        self.sdf_network, self.color_network = sdf_network, color_network
        self.set_precision(precision, fast_trig)
        self._key = None
        self._packs = {}
        self._stacks = None

    def __deepcopy__(self, memo):
        """A copy taken while a forward is in flight (an EMA snapshot from a callback) must not inherit the hold depth: nothing
        would ever release it, and the copy's version walk would be skipped for good (advisor, round 5).  Caches are not copied
        either -- they are keyed on the ORIGINAL parameters' addresses."""
        import copy
        twin = FieldPack.__new__(FieldPack)
        memo[id(self)] = twin
        twin.sdf_network = copy.deepcopy(self.sdf_network, memo)
        twin.color_network = copy.deepcopy(self.color_network, memo)
        twin.prec, twin.fast_trig = self.prec, self.fast_trig
        twin._key, twin._packs, twin._stacks = None, {}, None
        return twin

    def set_precision(self, precision, fast_trig=None):
        self.prec = _l.PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        self.fast_trig = (self.prec == _l.OI_PREC_BF16) if fast_trig is None else bool(fast_trig)
        self._key = None

    def _sds(self):
        # The module tree is static: walk it once (named_parameters() costs ~50 us per call, and this runs three times
        # per render).  In-place updates, load_state_dict and .to() keep the Parameter objects, so the (data_ptr,
        # _version) key below still sees every change; replacing a Parameter OBJECT needs `invalidate()`.
        cached = getattr(self, "_sd_cache", None)
        if cached is not None:
            return cached
        self._sd_cache = self._sds_walk()
        return self._sd_cache

    def invalidate(self):
        self._sd_cache, self._key, self._packs, self._plists, self._stacks = None, None, {}, None, None

    def param_lists(self):
        """(all parameters, style + FiLM-head parameters): cached flat lists for the requires-grad checks."""
        pl = getattr(self, "_plists", None)
        if pl is None:
            sd, csd = self._sds()
            named = list(sd.items()) + (list(csd.items()) if self.color_network is not None else [])
            film = [p for n, p in named if n.startswith("style.") or ".gamma." in n or ".beta." in n]
            pl = self._plists = ([p for _, p in named], film)
        return pl

    def _sds_walk(self):
        sd = dict(self.sdf_network.named_parameters())
        if self.color_network is not None:
            csd = dict(self.color_network.named_parameters())
        else:
            ref = sd["pts_linears.1.weight"]
            z = lambda *s: torch.zeros(*s, device=ref.device)
            csd = {"views_linears.weight": z(128, 131), "views_linears.bias": z(128), "rgb_linear.weight": z(3, 128),
                   "rgb_linear.bias": z(3), "views_linears.gamma.weight": z(128, 64), "views_linears.gamma.bias": z(128),
                   "views_linears.beta.weight": z(128, 64), "views_linears.beta.bias": z(128)}
        return sd, csd

    def _stack_cache(self):
        """params.StackCache, current: one gather launch per parameter version (CUDA); None on other devices."""
        sd, csd = self._refresh_key()
        if not sd["pts_linears.1.weight"].is_cuda:
            return None
        sc = getattr(self, "_stacks", None)
        if sc is None or sc.flat.device != sd["pts_linears.1.weight"].device:
            from .params import StackCache
            sc = self._stacks = StackCache(sd, csd)
            self._stacks_key = None
        if self._stacks_key != self._key:
            with torch.no_grad():
                sc.refresh()
            self._stacks_key = self._key
        return sc

    def stacked(self):
        sc = self._stack_cache()
        if sc is None:
            return stack_field_params(*self._sds())
        return sc.get(torch.is_grad_enabled())

    def hold(self, on):
        """Between hold(True) and hold(False) the caller guarantees that no parameter changes (one Generator.forward): the
        (data_ptr, _version) walk over ~65 parameters -- 20 us of host time, three times per render -- runs once.  A depth
        counter, not a flag: a nested forward that shares the pack (a render inside a render's callback) must not release the
        outer hold early, and the outermost release is what re-arms the version walk."""
        depth = getattr(self, "_held", 0)
        if on:
            if depth == 0:
                self._refresh_key()
            self._held = depth + 1
        else:
            self._held = max(0, depth - 1)

    def _refresh_key(self):
        sd, csd = self._sds()
        if getattr(self, "_held", False) and self._key is not None:
            return sd, csd
        key = tuple((p.data_ptr(), p._version) for p in list(sd.values()) + list(csd.values()))
        if key != self._key:
            self._packs = {}
            self._key = key
        return sd, csd

    def film_stacked(self, differentiable):
        """Stacked style / FiLM-head parameters for `oi_film_params`: views of the stack cache (no launch; the
        differentiable ones carry the graph back to the per-layer parameters)."""
        from .params import FILM_KEYS
        sc = self._stack_cache()
        if sc is not None:
            return sc.get(differentiable, FILM_KEYS)
        sd, csd = self._sds()
        if differentiable:
            return stack_field_params(sd, csd, keys=FILM_KEYS)
        if "film" not in self._packs:
            with torch.no_grad():
                self._packs["film"] = stack_field_params(sd, csd, keys=FILM_KEYS)
        return self._packs["film"]

    @property
    def prec_bwd(self):
        """The backward kernels exist for f32, f16x3 and bf16: BF16X6 differentiates through the exact-fp32 image, BF16X3
        through the F16X3 one (its own backward spilled 240 registers and was dropped in round 3)."""
        if self.prec == _l.OI_PREC_BF16X6:
            return _l.OI_PREC_F32
        return _l.OI_PREC_F16X3 if self.prec == _l.OI_PREC_BF16X3 else self.prec

    def packed(self, for_backward=False):
        sd, csd = self._refresh_key()
        prec = self.prec_bwd if for_backward else self.prec
        if prec not in self._packs:
            with torch.no_grad():
                sc = self._stack_cache()
                P = stack_field_params(sd, csd) if sc is None else sc.get(False)
                self._packs[prec] = ops.mlp_pack_weights(P["w0"], P["b0"], P["wh"], P["bh"], P["wsig"], P["bsig"],
                                                         P["wv"], P["bv"], P["wrgb"], P["brgb"], prec)
        return self._packs[prec]

    def check(self):
        """Finiteness report of the packed images (oi_mlp_pack_status): one stream sync, so call it where that is free --
        after a checkpoint load, before an inference run -- not inside a training iteration."""
        ops.mlp_pack_status(self.packed())

    def film(self, z=None, w=None):
        from .autograd import film_params
        return film_params(self, z=z, w=w)
