"""Fused optimiser steps and EMA update: one HIP launch per network per step.

`FusedAdam` / `FusedRMSprop` are drop-ins for the `torch.optim.Adam` / `torch.optim.RMSprop` instances the reference
builds from configs/train.yaml:133-147 (same constructor keywords, same `state_dict()` layout -- `step`,
`exp_avg`, `exp_avg_sq` / `square_avg` per parameter -- so optimiser checkpoints interchange), restricted to the
options that configuration uses (no weight decay, amsgrad, momentum, centering, maximize).  `ema_update` is the
parameter loop of `EMA.update` (src/utils/ema.py:26-32).  The kernels are `oi_multi_adam`, `oi_multi_rmsprop`,
`oi_multi_lerp` (csrc/optim.hip); there is no PyTorch fallback.
"""
import math

import numpy as np
import torch

from . import lib as _l


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _ChunkTable:
    """Device array of `oi_mt_chunk` descriptors, rebuilt only when a pointer changes (gradients dropped with
    `zero_grad(set_to_none=True)` come back at new addresses).  Uploads go through a small ring of pinned buffers with
    non-blocking copies: a pageable `.to(device)` would stall the host behind everything queued on the stream."""

    RING = 4

    def __init__(self):
        self.key, self.table, self.n = None, None, 0
        self._pinned, self._events, self._i = [], [], 0

    def _rows(self, key):
        """Descriptor rows of `key`.  The chunk offsets depend only on the element counts, so they are laid out once;
        when gradients come back at new addresses (every step with set_to_none=True) only the base pointers are
        re-broadcast -- a few numpy operations instead of a Python loop over ~700 chunks."""
        sizes = tuple(k[4] for k in key)
        lay = getattr(self, "_layout", None)
        if lay is None or lay[0] != sizes:
            chunk = _l.load().oi_mt_chunk_elems()
            counts = np.array([(n + chunk - 1) // chunk for n in sizes], dtype=np.int64)
            owner = np.repeat(np.arange(len(sizes)), counts)                      # parameter index of every row
            first = np.concatenate([[0], np.cumsum(counts)[:-1]])
            off = (np.arange(int(counts.sum())) - first[owner]) * chunk            # element offset inside the parameter
            m = np.minimum(chunk, np.array(sizes, dtype=np.int64)[owner] - off)
            lay = self._layout = (sizes, owner, off * 4, m)
        _, owner, off_b, m = lay
        ptr = np.array([k[:4] for k in key], dtype=np.int64)[owner]                # (rows, 4) base pointers
        arr = np.empty((owner.shape[0], 5), dtype=np.int64)
        arr[:, :4] = np.where(ptr != 0, ptr + off_b[:, None], 0)
        arr[:, 4] = m
        return arr

    def get(self, quads):
        """quads: list of (p, g, s0, s1) tensors (s0 / s1 may be None)."""
        key = tuple((p.data_ptr(), g.data_ptr(), 0 if a is None else a.data_ptr(), 0 if b is None else b.data_ptr(),
                     p.numel()) for p, g, a, b in quads)
        if key != self.key:
            arr = self._rows(key)  # 4 pointers + (n | reserved << 32): 40-byte structs
            n_rows = arr.shape[0]
            dev = quads[0][0].device
            if self.table is None or self.table.shape[0] < n_rows:
                self.table = torch.empty(max(n_rows, 1), 5, dtype=torch.int64, device=dev)
                self._pinned = [torch.empty(max(n_rows, 1), 5, dtype=torch.int64, pin_memory=True) for _ in range(self.RING)]
                self._events = [None] * self.RING
            i = self._i = (self._i + 1) % self.RING
            if self._events[i] is not None and not self._events[i].query():
                self._events[i].synchronize()
            self._pinned[i][:n_rows].numpy()[:] = arr
            self.table[:n_rows].copy_(self._pinned[i][:n_rows], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._events[i] = ev
            self.key, self.n = key, n_rows
        return self.table, self.n


def _written(tensors):
    """The kernels write through raw pointers, which autograd's version counters do not see.  Everything keyed on
    `Tensor._version` -- the packed MFMA weight images of `fields.FieldPack`, the cached light / variance scalars,
    autograd's saved-tensor checks -- must observe the update, so count it the way an in-place torch op would."""
    torch.autograd.graph.increment_version(tensors)


def _check(p):
    if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
        raise ValueError("fused optimisers need contiguous fp32 CUDA parameters")


class _StepCounts:
    """`state[p]["step"]` is what torch keeps -- a CPU scalar tensor per parameter, so optimiser checkpoints interchange --
    but a training step does not touch ~70 tensors for it (`+= 1` and `int()` per parameter: 0.3 ms of host time per
    generator step): the counts live in Python ints and are written into the state when somebody asks for it
    (`state_dict()`; `load_state_dict()` reads them back).  `opt.state[p]["step"]` read directly between steps is stale."""

    def _count(self, p, st):
        steps = self.__dict__.setdefault("_steps", {})
        k = steps.get(p)
        k = (int(st["step"]) if k is None else k) + 1
        steps[p] = k
        return k

    def _flush_steps(self):
        for p, k in self.__dict__.get("_steps", {}).items():
            self.state[p]["step"] = torch.tensor(float(k))

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__["_steps"] = {}

    # pickle / copy.deepcopy go through __getstate__ (torch.optim.Optimizer: defaults, state, param_groups): the counts are
    # written into the state first, and the copy starts from the state alone (a deep copy's parameters are new objects:
    # the Python-side dictionary keyed on the old ones would be stale, i.e. bias correction restarted at step 1).
    def __getstate__(self):
        self._flush_steps()
        return super().__getstate__()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_steps"] = {}
        self.__dict__.setdefault("_tables", {})


class FusedAdam(_StepCounts, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad are not used by configs/train.yaml")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _l.load()
        for gi, group in enumerate(self.param_groups):
            by_step = {}  # parameters that skipped steps (grad None) carry their own bias correction: one launch each
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check(p)
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                by_step.setdefault(self._count(p, st), []).append(
                    (p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            for k, (step, quads) in enumerate(sorted(by_step.items())):
                table, n = self._tables.setdefault((gi, k), _ChunkTable()).get(quads)
                rc = L.oi_multi_adam(table.data_ptr(), n, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                     1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step), _stream())
                if rc:
                    raise _l.OiHipError(L.oi_last_error().decode())
                _written([t for q in quads for t in (q[0], q[2], q[3])])
        return loss


class FusedRMSprop(_StepCounts, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False):
        if weight_decay != 0 or momentum != 0 or centered:
            raise NotImplementedError("FusedRMSprop: weight_decay / momentum / centered are not used by configs/train.yaml")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=0, momentum=0, centered=False))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _l.load()
        for gi, group in enumerate(self.param_groups):
            quads = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check(p)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                self._count(p, st)
                quads.append((p, p.grad.contiguous() if not p.grad.is_contiguous() else p.grad, st["square_avg"], None))
            if not quads:
                continue
            table, n = self._tables.setdefault(gi, _ChunkTable()).get(quads)
            rc = L.oi_multi_rmsprop(table.data_ptr(), n, float(group["lr"]), float(group["alpha"]), float(group["eps"]),
                                    _stream())
            if rc:
                raise _l.OiHipError(L.oi_last_error().decode())
            _written([t for q in quads for t in (q[0], q[2])])
        return loss


_EMA_TABLES = {}


@torch.no_grad()
def ema_update(ema_params, params, beta):
    """p_ema <- p.lerp(p_ema, beta) for every parameter pair, one launch (ema.py:26-30)."""
    pairs = [(pe, p, None, None) for pe, p in zip(ema_params, params)]
    if not pairs:
        return
    for pe, p, _, _ in pairs:
        _check(pe)
        _check(p)
    key = id(pairs[0][0])
    table, n = _EMA_TABLES.setdefault(key, _ChunkTable()).get(pairs)
    L = _l.load()
    rc = L.oi_multi_lerp(table.data_ptr(), n, float(beta), _stream())
    if rc:
        raise _l.OiHipError(L.oi_last_error().decode())
    _written([q[0] for q in pairs])
