"""Checkpoint IO in the reference's on-disk format (src/utils/checkpoint.py:16-137): ONE torch pickle
holding {module_key: state_dict, ...} for every registered module/optimizer/scheduler plus scalars
(`it`, `epoch`, `loss`, `ema@<beta>` ...), so `model.pt` files interchange with the reference in both
directions (the drop-in modules keep its state_dict keys).  The DDP 'module.' prefix is adapted on load
in either direction (checkpoint.py:109-137), for DistributedDataParallel and oi_amd.ddp.FlatGradDDP alike."""
import os

import torch
import torch.nn as nn


def _is_wrapped(m):
    return isinstance(m, nn.Module) and hasattr(m, "module") and isinstance(getattr(m, "module"), nn.Module) \
        and type(m).__name__ in ("DistributedDataParallel", "FlatGradDDP")


class CheckpointIO:
    def __init__(self, checkpoint_dir="./chkpts", **kwargs):
        self.module_dict = dict(kwargs)
        self.checkpoint_dir = checkpoint_dir
        if checkpoint_dir is not None:
            os.makedirs(checkpoint_dir, exist_ok=True)

    def register_modules(self, **kwargs):
        self.module_dict.update(kwargs)

    def save(self, filename, **scalars):
        if not os.path.isabs(filename):
            filename = os.path.join(self.checkpoint_dir, filename)
        out = dict(scalars)
        for k, v in self.module_dict.items():
            sd = v.state_dict()
            if type(v).__name__ == "FlatGradDDP":  # keep the reference's DDP key layout ('module.' prefix)
                sd = {kk: vv for kk, vv in sd.items() if kk.startswith("module.")}
            out[k] = sd
        torch.save(out, filename)
        return filename

    def load(self, filename, strict=True):
        sd = filename if isinstance(filename, dict) else torch.load(
            filename if os.path.isabs(filename) or os.path.exists(filename) else os.path.join(self.checkpoint_dir, filename),
            map_location="cpu", weights_only=False)
        return self.parse_state_dict(sd, strict=strict)

    def parse_state_dict(self, state_dict, strict=True):
        for k, v in self.module_dict.items():
            if k not in state_dict:
                if strict:
                    raise KeyError(f"{k} not found in checkpoint")
                continue
            part = state_dict[k]
            if isinstance(v, nn.Module):
                has_prefix = len(part) > 0 and next(iter(part.keys())).startswith("module.")
                target = v
                if _is_wrapped(v) and not has_prefix:
                    target = v.module
                elif not _is_wrapped(v) and has_prefix:
                    part = {kk[len("module."):]: vv for kk, vv in part.items()}
                elif _is_wrapped(v) and has_prefix and type(v).__name__ == "FlatGradDDP":
                    target, part = v.module, {kk[len("module."):]: vv for kk, vv in part.items()}
                target.load_state_dict(part, strict=strict)
            else:
                v.load_state_dict(part)
        return {k: v for k, v in state_dict.items() if k not in self.module_dict}
