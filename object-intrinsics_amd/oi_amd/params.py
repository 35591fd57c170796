"""Flat views of the reference-named parameters (SURVEY.md 8b state_dict keys) in the layouts the
C ABI expects (include/oi_hip.h).  Pure torch.stack/cat of small tensors: differentiable, so the
per-parameter gradients flow back to the reference-named nn.Parameters."""
import torch


FILM_KEYS = ("style_w", "style_b", "gw", "gb", "bw", "bb")


def stack_field_params(sd, csd, n_layers=8, keys=None):
    """sd: ShapeNetwork tensors by reference key; csd: ColorNetwork tensors by reference key.
    keys: optional subset of the output entries to build (each torch.stack is a device launch)."""
    L = n_layers
    g = lambda k: sd[k]
    if keys is not None:
        full = {
            "style_w": lambda: torch.stack([g(f"style.{i}.weight") for i in range(3)]),
            "style_b": lambda: torch.stack([g(f"style.{i}.bias") for i in range(3)]),
            "gw": lambda: torch.stack([g(f"pts_linears.{l}.gamma.weight") for l in range(L)] + [csd["views_linears.gamma.weight"]]),
            "gb": lambda: torch.stack([g(f"pts_linears.{l}.gamma.bias") for l in range(L)] + [csd["views_linears.gamma.bias"]]),
            "bw": lambda: torch.stack([g(f"pts_linears.{l}.beta.weight") for l in range(L)] + [csd["views_linears.beta.weight"]]),
            "bb": lambda: torch.stack([g(f"pts_linears.{l}.beta.bias") for l in range(L)] + [csd["views_linears.beta.bias"]]),
        }
        return {k: full[k]() for k in keys}
    out = {
        "style_w": torch.stack([g(f"style.{i}.weight") for i in range(3)]),
        "style_b": torch.stack([g(f"style.{i}.bias") for i in range(3)]),
        "w0": g("pts_linears.0.weight"),
        "b0": g("pts_linears.0.bias"),
        "wh": torch.stack([g(f"pts_linears.{l}.weight") for l in range(1, L)]),
        "bh": torch.stack([g(f"pts_linears.{l}.bias") for l in range(1, L)]),
        "wsig": g("sigma_linear.weight").reshape(-1),
        "bsig": g("sigma_linear.bias").reshape(-1),
        "wv": csd["views_linears.weight"],
        "bv": csd["views_linears.bias"],
        "wrgb": csd["rgb_linear.weight"],
        "brgb": csd["rgb_linear.bias"],
        "gw": torch.stack([g(f"pts_linears.{l}.gamma.weight") for l in range(L)] + [csd["views_linears.gamma.weight"]]),
        "gb": torch.stack([g(f"pts_linears.{l}.gamma.bias") for l in range(L)] + [csd["views_linears.gamma.bias"]]),
        "bw": torch.stack([g(f"pts_linears.{l}.beta.weight") for l in range(L)] + [csd["views_linears.beta.weight"]]),
        "bb": torch.stack([g(f"pts_linears.{l}.beta.bias") for l in range(L)] + [csd["views_linears.beta.bias"]]),
    }
    return out


def _stack_groups(sd, csd, n_layers=8):
    """name -> list of the tensors torch.stack would receive (the stacked entries of `stack_field_params`)."""
    L = n_layers
    g = lambda k: sd[k]
    film = lambda part, kind: [g(f"pts_linears.{l}.{part}.{kind}") for l in range(L)] + [csd[f"views_linears.{part}.{kind}"]]
    return {
        "style_w": [g(f"style.{i}.weight") for i in range(3)],
        "style_b": [g(f"style.{i}.bias") for i in range(3)],
        "wh": [g(f"pts_linears.{l}.weight") for l in range(1, L)],
        "bh": [g(f"pts_linears.{l}.bias") for l in range(1, L)],
        "gw": film("gamma", "weight"), "gb": film("gamma", "bias"), "bw": film("beta", "weight"), "bb": film("beta", "bias"),
    }


class _Stacked(torch.autograd.Function):
    """The stacked array as a differentiable function of its per-layer parameters WITHOUT a launch either way: the values
    already sit in the cache buffer (StackCache.refresh), the gradient of entry i is slice i of the incoming gradient."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.n = len(parts)
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[i] for i in range(ctx.n))


class StackCache:
    """Persistent stacked copies of the per-layer parameters (the layouts `oi_film_params` / `oi_mlp_pack_weights` take),
    refreshed by ONE gather launch (`oi_multi_copy`) when a parameter version changes -- `torch.stack` costs a launch per
    stacked array and call: 15 after every optimiser step and 15 more, with as many copies backward, per training render."""

    def __init__(self, sd, csd, n_layers=8):
        from .optim import _ChunkTable
        self.groups = _stack_groups(sd, csd, n_layers)
        ref = sd["pts_linears.1.weight"]
        offs, tot = {}, 0
        for k, ts in self.groups.items():
            if any(t.shape != ts[0].shape or t.dtype != torch.float32 for t in ts):
                raise ValueError(f"StackCache: the tensors of '{k}' differ in shape or are not fp32")
            offs[k] = tot
            tot += (len(ts) * ts[0].numel() + 3) // 4 * 4
        self.flat = torch.zeros(tot, dtype=torch.float32, device=ref.device)
        self.views = {k: self.flat[offs[k]:offs[k] + len(ts) * ts[0].numel()].view(len(ts), *ts[0].shape)
                      for k, ts in self.groups.items()}
        self._table = _ChunkTable()
        self.singles = {"w0": sd["pts_linears.0.weight"], "b0": sd["pts_linears.0.bias"], "wsig": sd["sigma_linear.weight"],
                        "bsig": sd["sigma_linear.bias"], "wv": csd["views_linears.weight"], "bv": csd["views_linears.bias"],
                        "wrgb": csd["rgb_linear.weight"], "brgb": csd["rgb_linear.bias"]}

    def __deepcopy__(self, memo):
        return None   # a copied module tree (EMA) builds its own cache over its own parameters (FieldPack._stack_cache)

    def refresh(self):
        from . import lib as _l
        from .optim import _stream
        quads = [(self.views[k][i], t.detach(), None, None) for k, ts in self.groups.items() for i, t in enumerate(ts)]
        for dst, src, _, _ in quads:
            if not src.is_cuda or not src.is_contiguous():
                raise ValueError("StackCache needs contiguous CUDA parameters")
        table, n = self._table.get(quads)
        _l.check(_l.load().oi_multi_copy(table.data_ptr(), n, _stream()), "oi_multi_copy")
        # a raw-pointer write: counted like an in-place op, so that a backward through a graph that saved the OLD contents
        # (forward, optimiser step, forward again, then backward of the first) fails loudly instead of using new values
        torch.autograd.graph.increment_version(self.flat)

    def get(self, differentiable, keys=None):
        """dict like `stack_field_params` (same keys and shapes).  differentiable: the stacked entries carry the graph back
        to their parameters (no launch), the single-tensor entries ARE the parameters."""
        out = {}
        for k in (keys if keys is not None else list(self.views) + list(self.singles)):
            if k in self.views:
                out[k] = _Stacked.apply(self.views[k], *self.groups[k]) if differentiable else self.views[k]
            else:
                t = self.singles[k]
                t = t.reshape(-1) if k in ("wsig", "bsig") else t
                out[k] = t if differentiable else t.detach()
        return out
