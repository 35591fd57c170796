"""RCCL tests of the data-parallel path on real GPUs: the REAL generator and both discriminators
wrapped in oi_amd.ddp.FlatGradDDP (communication stream on), driven by oi_amd.trainer.Trainer.  One rank runs on any GPU
box; the 2-rank test needs two GPUs and is skipped otherwise.  Each rank is its own process (one process per GPU)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "object-intrinsics_amd"))
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
wrap = os.environ["OI_WRAP"] == "1"
if wrap:
    dist.init_process_group("nccl", init_method="env://", device_id=dev)
from oi_amd.config import build_from_config
from oi_amd.ddp import FlatGradDDP
from oi_amd.optim import FusedAdam, FusedRMSprop
from oi_amd.trainer import Trainer
R = 16
torch.manual_seed(5); np.random.seed(5)          # same weights on every rank; FlatGradDDP broadcasts rank 0's anyway
gen, disc = bench.build_models(R, 8, 8, 1, "f16x3", dev)
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator",
                              aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1,
                              img_size=R, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).to(dev)
nets = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc}
if wrap:
    nets = {k: FlatGradDDP(v, comm_stream=True) for k, v in nets.items()}
mods = dict(nets)
mods["opt_generator"] = FusedAdam(nets["generator"].parameters(), lr=2e-5, betas=(0.0, 0.9))
mods["opt_discriminator"] = FusedRMSprop(nets["discriminator"].parameters(), lr=1e-4)
mods["opt_mask_discriminator"] = FusedRMSprop(nets["mask_discriminator"].parameters(), lr=1e-4)
tr = Trainer(mods)
# every rank sees the SAME data and RNG stream: the averaged gradient then equals the single-process gradient, so the
# wrapped runs (1 or 2 ranks) must reproduce the unwrapped run
g = torch.Generator(device=dev).manual_seed(9)
data = {"image": torch.rand(1, 3, R, R, device=dev, generator=g), "mask": torch.rand(1, 1, R, R, device=dev, generator=g)}
out = None
for step in range(3):
    torch.manual_seed(100 + step); np.random.seed(100 + step)
    out = tr.train_step(data)
torch.cuda.synchronize()
unwrap = lambda m: m.module if hasattr(m, "flat_grad") else m
res = {k: float(v) for k, v in out.items()}
res["gen_w"] = float(sum(p.double().sum() for p in unwrap(nets["generator"]).parameters()))
res["disc_w"] = float(sum(p.double().abs().sum() for p in unwrap(nets["discriminator"]).parameters()))
res["mdisc_w"] = float(sum(p.double().abs().sum() for p in unwrap(nets["mask_discriminator"]).parameters()))
if rank == 0:
    print("RESULT " + json.dumps(res), flush=True)
if wrap:
    dist.barrier(device_ids=[rank]); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, wrap):
    import json
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OI_WRAP="1" if wrap else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _close(a, b):
    for k in a:
        assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(b[k])), (k, a[k], b[k])


def test_flat_grad_ddp_single_rank_rccl_matches_unwrapped():
    """One RCCL rank: process group, parameter broadcast, flat views, communication-stream exchange + sync(), the
    deferred discriminator step of the trainer -- three training iterations give the losses / weights of the
    unwrapped run."""
    _close(_run(1, True), _run(1, False))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_flat_grad_ddp_two_ranks_rccl_matches_single_process():
    """Two RCCL ranks over xGMI fed identical data: the averaged gradients equal the single-process ones."""
    _close(_run(2, True), _run(1, False))
