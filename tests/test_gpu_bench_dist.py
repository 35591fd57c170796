"""The N > 1 path of bench.py, executed before the driver's multi-GPU run does: `python bench.py --gpus 2` through its own
launcher (torch.distributed.run, 127.0.0.1 rendezvous), two ranks on the one GPU of this box, exchanging through gloo
(RCCL cannot place two ranks on one device).  Everything else is the code the 8-GPU run executes: barriers, the MAX / SUM
reductions of the timing tensors, FlatGradDDP(comm_stream=True) under captured discriminator steps, the watchdog, the
single JSON line of rank 0 (scripts/train.py:50-56,136,157-158 is what this replaces)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env, train_steps=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--min-seconds", "0.1", "--train-steps", str(train_steps), "--no-extras", "--no-cpu-baseline", "--no-bf16"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]   # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_through_the_self_launcher():
    one = _run(1, {})
    two = _run(2, {"OI_BENCH_DIST_BACKEND": "gloo", "OI_BENCH_ONE_DEVICE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for line in (one, two):
        assert line["unit"] == "rays/s" and line["scaling"] == "weak" and line["steps"] == 3 and line["warmup"] == 1
        assert line["training"]["finite"] is True and line["training"]["steps"] == 2, line["training"]
        assert line["roofline"]["achieved"] and line["value"] > 0
    # whole-job value = units of all ranks / max-over-ranks time.  Two ranks SHARE one GPU here, so per-rank steps take
    # about twice as long and the aggregate stays near the single-rank rate: the arithmetic is what is checked --
    # value == 2 x 4096 rays x steps / (ms_per_step x steps)
    rays = two["config"]["rays_per_step_per_gpu"]
    assert abs(two["value"] - 2 * rays / (two["ms_per_step"] * 1e-3)) < 1e-6 * two["value"]
    assert abs(one["value"] - rays / (one["ms_per_step"] * 1e-3)) < 1e-6 * one["value"]
    assert 0.5 * one["value"] < two["value"] < 2.5 * one["value"], (one["value"], two["value"])
    assert "flat-gradient RCCL all-reduce x3" in two["training"]["what"]
    assert two["training"]["d_train_images_per_s"] == pytest.approx(4 * 2 * two["training"]["it_per_s"])
    assert isinstance(two["cpu_baseline"], str) and "N=1" in two["cpu_baseline"]
    assert two["dist_backend"].startswith("gloo")


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """The rank count of the driver's scaling run, before it: `python bench.py --gpus 8` -> eight processes (gloo, one device,
    the backward's working memory capped so that eight of them fit), one JSON line.  Checked: n_gpus, the whole-job arithmetic
    (8 x rays / max-over-ranks time), a finite training leg with the three flat-gradient exchanges, and the launcher's
    rank -> device map: LOCAL_RANK r would take cuda:r on an 8-GPU node (scripts/train.py:50-56 does the same)."""
    line = _run(8, {"OI_BENCH_DIST_BACKEND": "gloo", "OI_BENCH_ONE_DEVICE": "1", "OI_BWD_SCRATCH_MB": "1024"})
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    rays = line["config"]["rays_per_step_per_gpu"]
    assert abs(line["value"] - 8 * rays / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["training"]["finite"] is True and "flat-gradient RCCL all-reduce x3" in line["training"]["what"]
    assert line["training"]["rays_per_s"] == pytest.approx(3 * 8 * rays * line["training"]["it_per_s"])
    rows = sorted(line["rank_devices"]["rows"])
    assert [r[0] for r in rows] == list(range(8)) and all(r[1] == r[0] == r[2] for r in rows), rows
    assert all(r[3] == 0 for r in rows)   # (this rehearsal: every rank on the one GPU)
    assert "dp8" in line["config"]["parallelism"]
