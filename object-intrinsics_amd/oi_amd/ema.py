"""Parameter EMA of the generator (src/utils/ema.py:7-41): p_ema <- lerp(p, p_ema, beta), buffers copied.
The parameter loop is ONE launch of `oi_multi_lerp` (oi_amd.optim.ema_update, csrc/optim.hip) instead of one lerp + copy
per parameter.  Like every op of the path there is no host-tensor branch: parameters that are not on the GPU raise."""
import copy

import torch
import torch.nn as nn


class EMA:
    def __init__(self, m, beta, m_ema=None):
        if hasattr(m, "module") and isinstance(m.module, nn.Module) and type(m).__name__ in ("DistributedDataParallel", "FlatGradDDP"):
            m = m.module
        if m_ema is None:
            m_ema = copy.deepcopy(m)
        m_ema = m_ema.eval()
        for p in m_ema.parameters():
            p.requires_grad = False
        self.m, self.m_ema, self.beta = m, m_ema, beta

    @property
    def module(self):
        return self.m_ema

    @torch.no_grad()
    def update(self, it=None):
        pe, p = list(self.m_ema.parameters()), [q.detach() for q in self.m.parameters()]
        if pe:
            from .optim import ema_update   # p.lerp(p_ema, beta) = p + beta (p_ema - p); rejects host tensors
            ema_update(pe, p, self.beta)
        if hasattr(self.m, "sync_it"):  # Generator keeps its iteration counter on the host between observations
            self.m.sync_it()
        for b_ema, b in zip(self.m_ema.buffers(), self.m.buffers()):
            b_ema.copy_(b)

    def get_state_dict(self):
        return {"state_dict": self.m_ema.state_dict(), "beta": self.beta}

    def __str__(self):
        return f"ema@{self.beta}"
