"""Data parallelism for the path: one process per GPU, weights replicated, ONE
all-reduce(sum)/world per network per backward on a flat fp32 gradient buffer, issued by RCCL over
xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

Replaces torch.nn.parallel.DistributedDataParallel as used by the reference (scripts/train.py:50-56,
157-158; tu/ddp.py) with the minimum the workload needs:
  * gradients of a network live in ONE contiguous buffer (param.grad are views into it), so the exchange
    is a single collective of 1.2 MB (generator) / 11 MB (each discriminator) instead of bucketed copies;
    at these sizes the ring is latency-bound, so fewer, larger messages are what pays on xGMI;
  * the collective is enqueued from an end-of-backward callback (same trigger DDP uses), stream-ordered
    before the optimiser step: drop-in for `loss.backward(); opt.step()` in the unmodified trainer;
  * parameters are broadcast once from rank 0 at construction; the reference's per-forward buffer
    broadcast (generator `it` + camera matrices, 3x per step) is dropped: those buffers are
    deterministic functions of the step counter / config and identical on every rank.
`.module` gives the wrapped network, as with DistributedDataParallel (the trainer uses it)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatGradDDP(nn.Module):
    def __init__(self, module, process_group=None, broadcast_parameters=True):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters()]
        self._params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._pending = False
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        if broadcast_parameters and self.world > 1:
            flat = torch.cat([p.detach().reshape(-1) for p in params])
            dist.broadcast(flat, src=0, group=self.pg)
            off = 0
            with torch.no_grad():
                for p in params:
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # -- gradient exchange -------------------------------------------------------------------
    def _on_grad(self, p):
        # autograd may have replaced p.grad with a fresh tensor (first accumulation into a None grad): fold it back
        if p.grad is not None and p.grad.data_ptr() != self._view_of(p).data_ptr():
            v = self._view_of(p)
            v.add_(p.grad) if getattr(self, "_accumulating", False) else v.copy_(p.grad)
            p.grad = v
        if not self._pending:
            self._pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _view_of(self, p):
        if not hasattr(self, "_views"):
            self._views = {}
            off = 0
            for q in self._params:
                self._views[id(q)] = self.flat_grad[off:off + q.numel()].view_as(q)
                off += q.numel()
        return self._views[id(p)]

    def _finalize(self):
        self._pending = False
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_grad.div_(self.world)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p in self._params:
            p.grad = self._view_of(p)


def zero_grad(optimizer_or_module):
    """`opt.zero_grad()` replacement that keeps the flat views (set_to_none=False)."""
    if isinstance(optimizer_or_module, torch.optim.Optimizer):
        optimizer_or_module.zero_grad(set_to_none=False)
    else:
        optimizer_or_module.zero_grad(set_to_none=False)
