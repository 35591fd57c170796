"""world_size-2 gloo test of the flat-gradient data-parallel wrapper (runs on CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "object-intrinsics_amd"))
    from oi_amd.ddp import FlatGradDDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank: construction must broadcast rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    ddp = FlatGradDDP(net)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    torch.manual_seed(7)
    xs = torch.randn(world, 4, 5)  # same on every rank; each rank takes its shard
    for step in range(2):
        opt.zero_grad(set_to_none=False)
        loss = ddp(xs[rank]).pow(2).sum()
        loss.backward()
        g = ddp.flat_grad.clone()
        opt.step()
    w1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, w0.numpy(), g.numpy(), w1.numpy()))
    dist.destroy_process_group()


def test_flat_grad_ddp_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0a, ga, w1a), (_, w0b, gb, w1b) = [(r, *[torch.from_numpy(a) for a in rest]) for r, *rest in res]
    assert torch.equal(w0a, w0b), "parameters were not broadcast from rank 0"
    assert torch.allclose(ga, gb) and torch.allclose(w1a, w1b), "ranks diverged"
    # reference: single process, mean of the two shards' gradients
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    torch.manual_seed(7)
    xs = torch.randn(world, 4, 5)
    for step in range(2):
        opt.zero_grad()
        loss = sum(net(xs[r]).pow(2).sum() for r in range(world)) / world
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        opt.step()
    w1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(g, ga, atol=1e-6) and torch.allclose(w1, w1a, atol=1e-6)


def _worker_asym(rank, world, port, q):
    """Rank 1's backward produces NO gradient for the wrapped network: `sync()` must still take part in the collective."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "object-intrinsics_amd"))
    from oi_amd.ddp import FlatGradDDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    net = torch.nn.Linear(4, 2)
    other = torch.nn.Linear(4, 2)
    ddp = FlatGradDDP(net)
    x = torch.ones(3, 4)
    ddp.zero_grad()
    loss = ddp(x).sum() if rank == 0 else other(x).sum()
    loss.backward()
    ddp.sync()
    q.put((rank, ddp.flat_grad.clone().numpy()))
    dist.destroy_process_group()


def test_flat_grad_ddp_rank_without_gradient_does_not_hang():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_asym, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, g1 = torch.from_numpy(res[0]), torch.from_numpy(res[1])
    assert torch.equal(g0, g1)
    # d sum(Wx + b) / dW = 3 (three rows of ones), / db = 3; averaged with rank 1's zeros
    assert torch.allclose(g0, torch.full_like(g0, 1.5))
