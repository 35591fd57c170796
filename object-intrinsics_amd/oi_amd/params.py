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
