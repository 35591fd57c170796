"""Data parallelism for the path: one process per GPU, weights replicated, ONE
all-reduce(mean) per network per backward on a flat fp32 gradient buffer, issued by RCCL over xGMI (torch.distributed
backend "nccl" is RCCL on ROCm; ReduceOp.AVG: the division happens inside the collective, no scale launch behind it).

Replaces torch.nn.parallel.DistributedDataParallel as used by the reference (scripts/train.py:50-56,
157-158; tu/ddp.py) with the minimum the workload needs:
  * gradients of a network live in ONE contiguous buffer (param.grad are views into it), so the exchange
    is a single collective of 1.2 MB (generator) / 11 MB (each discriminator) instead of bucketed copies;
    at these sizes the ring is latency-bound, so fewer, larger messages are what pays on xGMI;
  * the collective is enqueued from an end-of-backward callback (same trigger DDP uses), stream-ordered before
    the optimiser step: drop-in for `loss.backward(); opt.step()` in the unmodified trainer.  With
    `comm_stream=True` it runs on a dedicated communication stream that first waits for the backward's kernels, and
    `wait()` / `sync()` make the current stream wait for it: a trainer that knows what comes next can enqueue
    independent work (the next no-grad render) between `backward()` and `sync(); opt.step()` and have it overlap
    the exchange (oi_amd.trainer does);
  * `sync()` is the explicit form for callers that cannot rely on the hook: it issues the collective if this
    backward did not (a rank on which no wrapped parameter received a gradient would otherwise skip the
    collective and hang the others) -- call it on every rank after every backward of this network;
  * parameters are broadcast once from rank 0 at construction; the reference's per-forward buffer
    broadcast (generator `it` + camera matrices, 3x per step) is dropped: those buffers are
    deterministic functions of the step counter / config and identical on every rank.
Whenever a process group exists the broadcast and the collectives are ISSUED, also in a group of one rank (a 1-GPU box
then launches the same RCCL kernels on the communication stream that N ranks do); without a process group the wrapper
only keeps the flat gradient views.  `exchange_enabled = False` (tests) makes `_exchange()` a no-op that leaves this
rank's own gradients in the buffer.
`.module` gives the wrapped network, as with DistributedDataParallel (the trainer uses it)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatGradDDP(nn.Module):
    def __init__(self, module, process_group=None, broadcast_parameters=True, comm_stream=False):
        super().__init__()
        self.module = module
        self.pg = process_group
        self._dist = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self._dist else 1
        self.exchange_enabled = True
        self._avg = self._dist and dist.get_backend(process_group) == "nccl"   # RCCL averages inside the reduction
        params = [p for p in module.parameters()]
        self._params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._views = {}
        off = 0
        for p in params:
            v = self.flat_grad[off:off + p.numel()].view_as(p)
            self._views[id(p)] = v
            p.grad = v
            off += p.numel()
        self._pending = False      # an end-of-backward callback is queued
        self._needs_exchange = True  # set by zero_grad() / an arriving gradient, cleared by the collective
        self._event = None         # completion of the collective on the communication stream
        self._stream = torch.cuda.Stream(device=dev) if (comm_stream and dev.type == "cuda") else None
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        if broadcast_parameters and self._dist:
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
        if getattr(self, "_in_graph", False):   # inside GraphedDStep._step: the hook would run at capture time only;
            return                               # the captured step exchanges explicitly right after its replay
        # autograd may have replaced p.grad with a fresh tensor (first accumulation into a None grad): fold it back
        v = self._views[id(p)]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        self._needs_exchange = True
        if not self._pending:
            self._pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _finalize(self):
        self._pending = False
        self._exchange()

    def _exchange(self):
        self._needs_exchange = False
        if not self._dist or not self.exchange_enabled:
            return
        if self._stream is None:
            self._all_reduce_mean()
            return
        cur = torch.cuda.current_stream(self.flat_grad.device)
        self._stream.wait_stream(cur)                       # the backward's kernels first
        with torch.cuda.stream(self._stream):
            self._all_reduce_mean()
            self._event = torch.cuda.Event()
            self._event.record(self._stream)

    def _all_reduce_mean(self):
        """Mean over the ranks of the flat gradient buffer: ONE collective.  RCCL divides inside the reduction
        (ReduceOp.AVG); gloo (the CPU tests, the one-GPU rehearsals) has no AVG and takes SUM + one scale launch."""
        if self._avg:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG, group=self.pg)
            return
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        if self.world > 1:
            self.flat_grad.mul_(1.0 / self.world)

    def sync(self):
        """Issue the collective now unless this round's end-of-backward callback already did (a round starts at
        zero_grad(): a rank whose backward produced no gradient for this network still takes part, with zeros), then
        `wait()`."""
        if self._needs_exchange:
            self._exchange()
        self.wait()

    def wait(self):
        """Make the current stream wait for the exchange (call before reading the gradients / the optimiser step)."""
        if self._event is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_event(self._event)
            self._event = None

    def zero_grad(self, set_to_none=False):
        self.wait()
        self.flat_grad.zero_()
        self._needs_exchange = True
        for p in self._params:
            p.grad = self._views[id(p)]


def zero_grad(optimizer_or_module):
    """`opt.zero_grad()` replacement that keeps the flat views (set_to_none=False)."""
    optimizer_or_module.zero_grad(set_to_none=False)
