#!/usr/bin/env python3
"""Benchmark of the object-intrinsics hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W [--precision f16x3|bf16x6|f32|bf16x3|bf16] [--batch B]

Metric (BASELINE.json): rendered rays/sec at a 64x64 crop with 128 samples/ray (64 coarse + 64
importance, 1 up-sampling step = BASELINE config C2), plus discriminator images/sec, whole job
over N GPUs (weak scaling: every rank renders its own batch; the forward path has no collective).

One "step" = one `Generator.forward` (pose/latent sampling -> rays -> hierarchical sampling ->
FiLM-SIREN MLP with analytic normals + albedo -> compositing + Phong maps) over `--batch` images
per GPU with inputs generated on the device, followed by one ADA-discriminator forward
(`ADADiscriminatorView`, 64^2) on the rendered images.  Timing: W warm-up steps, then exactly K
steps bracketed by barrier + torch.cuda.synchronize(), max over ranks; rank 0 prints ONE JSON line.

`--gpus N` with N > 1 launched from a bare shell (no WORLD_SIZE in the environment) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU over RCCL.
The timed region is repeated (each repeat = exactly K steps between barrier + synchronize) until >= 2 s have been
measured; `value` / `ms_per_step` are over all repeats, `repeats` and the per-repeat spread are reported.

Extra objects in that line:
  roofline      dominant kernel (sdf_mlp_full3_kernel: sdf + d sdf/dx + albedo): algorithmic FLOPs / launch divided by
                its mean duration measured with HIP events on the launch stream inside the timed region
  cpu_baseline  the oracle (CPU restatement, oracle/oi_oracle.py) timed on the host cores on the same
                workload (N=1, rank 0 only) -- a reported baseline, not the target
"""
import argparse
import contextlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "object-intrinsics_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic FLOPs per point (GEMM MACs x 2 only), SURVEY.md 8(d) / BASELINE.md 4
F_SDF, F_GRAD, F_COL = 230400, 230400, 34304
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "bf16x6": 2500.0, "f16x3": 2500.0}  # MI355X_MICROARCH.md (dense)
MFMA_PER_MAC = {"f32": 1, "bf16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}
NET_KW = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
SDF_NPZ = os.path.join(ROOT, "tests", "golden", "weights_sdf.npz")


def example_cfg(R):
    """data/example/cfg.yaml + scripts/train.py:25-47, 88-115."""
    fov, img, img_scene = 10.0, 256, 1588
    cam_dist = float(1 / np.tan(0.5 * fov * np.pi / 180))
    scene_fov = float(2 * np.arctan(img_scene / img * np.tan(0.5 * fov * np.pi / 180)) * 180 / np.pi)
    return cam_dist, scene_fov, int(R * img_scene / img)


def build_models(R, S, I, K, precision, device):
    from oi_amd.config import build_from_config
    cam_dist, scene_fov, scene_res = example_cfg(R)
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    gen = build_from_config(net(
        "src.models.generator.Generator",
        color_network=net("src.models.fields.ColorNetwork", **NET_KW),
        sdf_network=net("src.models.fields.ShapeNetwork", checkpoint_path=SDF_NPZ, **NET_KW),
        deviation_network=net("src.third_party.neus.models.fields.SingleVarianceNetwork", init_val=0.3),
        light_network=net("src.utils.prior.build_directional_light_optimizable", cam_loc=None, light_loc=None,
                          ambient_color=0.33, diffuse_color=0.66, specular_color=0, shininess=10),
        camera=net("src.models.camera_network.Camera", cam_dist=cam_dist, resolution=scene_res, fov=scene_fov),
        z_dim=64, resolution=R, scene_resolution=scene_res,
        renderer=net("src.third_party.neus.models.renderer.NeuSRenderer", n_importance=I, n_outside=0, n_samples=S,
                     perturb=1, up_sample_steps=K),
        anneal_end=50000,
        pose_prior=net("src.utils.pose_sampler.Plane", cam_loc=[0, -1, 0], rot_degree_range_scale=360,
                       rot_roll_degree_range_scale=20, xy_range_scale=[6, 3.5])))
    gen.renderer.pack.set_precision(precision)
    disc = build_from_config(net(
        "src.models.discriminator.ADADiscriminatorView",
        aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=R, in_dim=3,
        last_bias=False, n_feat=512, out_dim=7, out_dim_latent=0, out_dim_position=6))
    return gen.to(device), disc.to(device)


class KernelTimer:
    """HIP events around selected launches on the current stream (where oi_amd launches)."""

    # An event pair around a launch costs that launch two marker packets: 5.7 us of idle stream in front of the kernel and 5.6 us
    # behind it (rocprofv3 timeline of the step) -- 1 % of a C2 step, 2 % in the bf16 mode, charged to the very region the events
    # are there to describe.  So the pairs go around every EVERY-th launch of the timed region (hundreds of samples per run;
    # `kernel_ms_samples` in the line), not around each.
    EVERY = 4

    def __init__(self):
        self.pairs = []
        self.calls = 0

    def due(self):
        self.calls += 1
        return self.calls % self.EVERY == 1   # (the first launch of a region is always sampled)

    def wrap(self, fn):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        return inner

    def reset(self):
        self.pairs = []

    def mean_ms(self):
        if getattr(self, "override", None) is not None:
            return self.override
        return float(np.mean([a.elapsed_time(b) for a, b in self.pairs])) if self.pairs else None


def cpu_baseline(R, S, I, K, B):
    """The oracle on the host cores: same workload (one B x R x R image, S+I samples/ray), forward."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oi_oracle as O
    torch.manual_seed(0)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(SDF_NPZ).items()}
    csd = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "weights_color.npz")).items()}
    # torch-CPU elementwise/GEMM ops of this size stop scaling (and collapse from oversubscription) beyond
    # ~32 threads: measured on the 256-core GPU-box host, 16/32/64/256 threads -> 564/576/401/26 rays/s
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    N = B * R * R
    g = torch.Generator().manual_seed(0)
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3) + 0.05 * torch.randn(N, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.1 * torch.randn(N, 3, generator=g), dim=-1)
    near, far = O.near_far_from_sphere(ro, rd)
    w = O.style_mlp(sd, torch.randn(B, 64, generator=g))
    lsd = {"param_direction": torch.tensor([0.0, 0.0, -1.0]), "param_ambient": torch.tensor(-0.7),
           "param_specular": torch.tensor(0.0), "param_shininess": torch.tensor(10.0)}
    w2b = torch.eye(4).repeat(B, 1, 1)
    # bounded sample: whole images until ~12 s of CPU work have been spent (at least 2, at most 8 images)
    times = []
    with torch.no_grad():
        while len(times) < 2 or (sum(times) < 12.0 and len(times) < 8):
            t0 = time.time()
            out = O.render(sd, csd, torch.tensor(0.3), ro, rd, near, far, w, S, I, K, 0.0)
            O.render_maps(out, ro, lsd, w2b, torch.rand(B, 3), B, R, R)
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": N / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full {B}x{R}x{R} images, {S}+{I} samples/ray, forward render + maps, mean of "
                      f"{len(times)} runs ({sum(times):.1f} s of CPU work, torch CPU fp32, {cores} threads)"}


def bench_training(args, gen, disc, device, world, barrier, distributed, sub_legs=True):
    from oi_amd.config import build_from_config
    from oi_amd.ddp import FlatGradDDP
    from oi_amd.trainer import Trainer
    R, B = args.res, args.batch
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    mdisc = build_from_config(net("src.models.discriminator.ADADiscriminator",
                                  aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1,
                                  img_size=R, in_dim=1, last_bias=False, n_feat=512, out_dim=1)).to(device)
    nets = {"generator": gen, "discriminator": disc, "mask_discriminator": mdisc}
    if distributed:
        nets = {k: FlatGradDDP(v, comm_stream=True) for k, v in nets.items()}
    mods = dict(nets)
    from oi_amd.optim import FusedAdam, FusedRMSprop  # configs/train.yaml:133-147, one launch per step each
    mods["opt_generator"] = FusedAdam(nets["generator"].parameters(), lr=2e-5, betas=(0.0, 0.9))
    mods["opt_discriminator"] = FusedRMSprop(nets["discriminator"].parameters(), lr=1e-4)
    mods["opt_mask_discriminator"] = FusedRMSprop(nets["mask_discriminator"].parameters(), lr=1e-4)
    tr = Trainer(mods, graph_d_steps=not getattr(args, "eager_d_steps", False))
    data = {"image": torch.rand(B, 3, R, R, device=device), "mask": torch.rand(B, 1, R, R, device=device)}
    for _ in range(2):
        tr.train_step(data)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        out = tr.train_step(data)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    it_s = args.train_steps / dt

    if not sub_legs:
        return {"it_per_s": it_s, "ms_per_it": 1e3 / it_s, "steps": args.train_steps,
                "finite": bool(all(torch.isfinite(torch.as_tensor(v)).all() for v in out.values()))}

    # the training render alone (forward + backward incl. the double-backward through the
    # normals: image, mask and eikonal terms all carry gradient) and one discriminator training step alone
    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    g_raw = gen
    from oi_amd.trainer import toggle_grad

    def render_fwd_bwd():
        toggle_grad(g_raw, True)
        for p_ in g_raw.parameters():
            p_.grad = None
        blob = g_raw(bs=B, it=tr.it, data={})["box"]
        ro = blob["render_out"]
        (ro["image"].square().mean() + ro["mask"].mean() + 10.0 * blob["loss"]["eikonal"]).backward()

    n_sub = max(3, min(10, args.train_steps))
    t_render = timed(render_fwd_bwd, n_sub)
    if sub_legs == "render":   # (the bf16-mode leg: iteration rate + the training render alone)
        return {"it_per_s": it_s, "ms_per_it": 1e3 / it_s, "steps": args.train_steps,
                "rays_per_s": 3 * world * B * R * R * it_s,
                "render_fwd_bwd": {"ms": 1e3 * t_render, "rays_per_s_per_gpu": B * R * R / t_render},
                "finite": bool(all(torch.isfinite(torch.as_tensor(v)).all() for v in out.values()))}
    with torch.no_grad():
        fake = g_raw(bs=B, it=tr.it, data={})["box"]
    fake_d = {**fake["render_out"], "c2b": fake["prior_info"]["c2b"]}
    t_dstep = timed(lambda: tr.train_step_discriminator("discriminator", data, fake_d), n_sub)
    tr_eager = Trainer(mods)
    tr_eager.it = tr.it
    # the eager comparison runs on the stream the captured step was recorded on: autograd pinned the discriminator's
    # AccumulateGrad nodes to it, and a backward from another stream warns about the mismatch
    gd = (tr._graphed or {}).get("discriminator")
    cap = getattr(gd, "stream", None)
    if cap is not None:
        cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap) if cap is not None else contextlib.nullcontext():
        t_dstep_eager = timed(lambda: tr_eager.train_step_discriminator("discriminator", data, fake_d), n_sub)
    if cap is not None:
        torch.cuda.current_stream().wait_stream(cap)
    return {"it_per_s": it_s, "ms_per_it": 1e3 / it_s, "steps": args.train_steps,
            "rays_per_s": 3 * world * B * R * R * it_s, "d_train_images_per_s": 4 * world * B * it_s,
            "render_fwd_bwd": {"ms": 1e3 * t_render, "rays_per_s_per_gpu": B * R * R / t_render,
                               "what": "Generator.forward with gradient + backward (double-backward through d sdf/dx)"},
            "d_step": {"ms": 1e3 * t_dstep, "images_per_s_per_gpu": 2 * B / t_dstep, "eager_ms": 1e3 * t_dstep_eager,
                       "what": "one ADADiscriminatorView training step: real fwd + R1 double-backward + fake fwd + bwd + RMSprop"
                               + (" (forward + backward replayed from one captured hipGraph; eager_ms = the same step "
                                  "launch by launch)" if not getattr(args, "eager_d_steps", False) else "")},
            "what": "Trainer.train_step: G step (render fwd+bwd incl. double-backward, 2 D fwd+bwd-to-input) + D step "
                    "+ mask-D step (each: real fwd + R1 double-backward + fake fwd + bwd), fused Adam/RMSprop steps"
                    + (", flat-gradient RCCL all-reduce x3" if distributed else ""),
            "finite": bool(all(torch.isfinite(torch.as_tensor(v)).all() for v in out.values()))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16x6", "f16x3", "bf16x3", "bf16"],
                    help="MFMA operand mode of the MLP contractions. bf16x6 (default) and f32 are the fp32-exact 1e-4 "
                         "parity paths (same tolerances in tests/); bf16x3 ~2e-4; bf16 ~1e-2")
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (training.batch_size: 1)")
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--importance", type=int, default=64)
    ap.add_argument("--up-steps", type=int, default=1)
    ap.add_argument("--train-steps", type=int, default=20, help="full GAN training iterations timed after the main region (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-mode measurement")
    ap.add_argument("--train-timeout", type=int, default=300, help="watchdog (s) for the secondary training leg")
    ap.add_argument("--no-disc", action="store_true")
    ap.add_argument("--eager-d-steps", action="store_true", help="training legs: discriminator steps eagerly instead of from captured hipGraphs")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step timed region until this much has been measured")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (C5, D batch sweep, shipped-config training, inference)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (one process per GPU, RCCL over xGMI)
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal switches for a 1-GPU box (tests/test_gpu_bench_dist.py): OI_BENCH_ONE_DEVICE=1 puts every rank on cuda:0,
    # OI_BENCH_DIST_BACKEND=gloo exchanges through the host (RCCL cannot place two ranks on one device).  Everything else
    # of the N > 1 path -- self-launch, rendezvous, barriers, MAX / SUM reductions, FlatGradDDP on a communication stream,
    # captured D steps, the watchdog -- is the code the multi-GPU run executes.
    backend = os.environ.get("OI_BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("OI_BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # OI_BENCH_FORCE_DIST=1 runs the RCCL code paths (process group, barrier, max/sum reductions, FlatGradDDP
    # all-reduce) with a single rank: a smoke test of the N > 1 path on a 1-GPU box.
    distributed = world > 1 or os.environ.get("OI_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=device)
        else:
            dist.init_process_group(backend, init_method="env://")
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"

    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)  # scripts/train.py:136: seed + rank
    R, S, I, K, B = args.res, args.samples, args.importance, args.up_steps, args.batch
    gen, disc = build_models(R, S, I, K, args.precision, device)
    gen.train()   # perturb on: per-ray jitter drawn on the device, random poses / latents / bg per step
    disc.eval()

    import oi_amd.autograd as A
    timer = KernelTimer()
    orig = A.sdf_mlp

    def sdf_mlp_timed(pack, pts, gamma, beta, B_, want_grad, want_rgb, want_feat, scratch=None, **kw):
        fn = timer.wrap(orig) if (want_grad and timer_on[0] and timer.due()) else orig
        return fn(pack, pts, gamma, beta, B_, want_grad, want_rgb, want_feat, scratch, **kw)

    timer_on = [False]
    A.sdf_mlp = sdf_mlp_timed
    import oi_amd.renderer as RR
    RR.sdf_mlp = sdf_mlp_timed

    def step(it):
        with torch.no_grad():
            out = gen(bs=B, it=it, data={})["box"]["render_out"]
            d = None if args.no_disc else disc(out["image"].contiguous(), it=it)
        return out, d

    def dist_barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    def barrier():
        torch.cuda.synchronize()  # (also in front: a host-side backend's barrier does not wait for this rank's stream)
        if distributed:
            dist_barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    # discriminator-only timing (images/s), untimed with respect to the main region
    d_img_s = d_img_s_eager = None
    if not args.no_disc:
        from oi_amd.graphed import GraphedDForward
        x = torch.rand(B, 3, R, R, device=device)
        barrier()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(args.steps):
                disc(x, it=0)
        barrier()
        d_img_s_eager = B * args.steps / (time.perf_counter() - t0)
        # the same forward through GraphedDForward (batch <= 4: the library's plan; else a captured hipGraph); augmentation
        # parameters still drawn per call on the host
        gd = GraphedDForward(disc)
        for _ in range(3):
            gd(x)
        n_d = max(args.steps, 100)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_d):
            gd(x)
        barrier()
        d_img_s = B * n_d / (time.perf_counter() - t0)

    # how many repeats of the K-step region make >= --min-seconds (same number on every rank: decided from the slowest)
    barrier()
    t0 = time.perf_counter()
    for i in range(min(5, args.steps)):
        step(args.warmup + i)
    barrier()
    probe = torch.tensor([(time.perf_counter() - t0) / min(5, args.steps)], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(probe, op=dist.ReduceOp.MAX)
    repeats = int(max(1, min(500, math.ceil(1.3 * args.min_seconds / max(1e-6, float(probe) * args.steps)))))
    timer_on[0] = True
    dts = []
    for r_ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        dts.append(time.perf_counter() - t0)
    timer_on[0] = False
    main_kernel_ms, args._kernel_samples = timer.mean_ms(), len(timer.pairs)
    timer.pairs = []  # the events are no longer needed (hundreds of them after the repeats)

    # ---- secondary: the same step in the bf16 operand mode that BASELINE.json's configs[1] names (one bf16 MFMA per
    #      product, v_sin/v_cos, fp16 scratch: 1e-2-class, NOT the parity path) -- reported beside the headline
    bf16_mode = None
    if world == 1 and args.precision != "bf16" and not args.no_bf16:
        try:
            gen.renderer.pack.set_precision("bf16")
            for i in range(3):
                step(i)
            timer.reset()
            timer.calls = 0
            timer_on[0] = True
            torch.cuda.synchronize()
            n_b = max(args.steps, 80)   # (its own step count: short steps, and the event pairs sample every 4th launch)
            t1 = time.perf_counter()
            for i in range(n_b):
                step(args.warmup + i)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t1
            timer_on[0] = False
            kms, kms_n = timer.mean_ms(), len(timer.pairs)
            fl = B * R * R * (args.samples + args.importance) * (F_SDF + F_GRAD + F_COL)
            ach = fl / (kms * 1e-3) / 1e12 if kms else None
            tr_b, tr_src = measured_traffic(args, "bf16")
            bf16_mode = {"value": B * R * R * n_b / dtb, "unit": "rays/s", "ms_per_step": dtb / n_b * 1e3, "steps": n_b,
                         "kernel_ms": kms, "kernel_ms_samples": kms_n,
                         "roofline": {"bound": "mfma", "kernel": "film_images_b_kernel + sdf_mlp_full3p_kernel (per-element bf16 images diag(gamma) W built per call, "
                                      "then the register-resident kernel: one bf16 MFMA per product, accumulator = phase, cosines parked as "
                                      "fp16 pairs in AGPRs, no scratch stream; kernel_ms covers BOTH launches)", "achieved": ach, "peak": PEAK_TFLOPS["bf16"],
                                      "unit": "TFLOP/s", "frac": (ach / PEAK_TFLOPS["bf16"]) if ach else None,
                                      "traffic": tr_b, "traffic_source": tr_src,
                                      "hbm": ({"achieved": tr_b / (kms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                               "frac": tr_b / (kms * 1e-3) / 8e12} if (kms and tr_b) else None)},
                         "note": "same workload with --precision bf16 (BASELINE configs[1]); tolerances on the rendered maps "
                         "measured against the reference's F4 / F5 outputs and the fp32 oracle at the full C2 size: image 5e-3, "
                         "mask 6e-3, colour 2.5e-3 worst pixel (tests/test_gpu_modules.py BF16_*_TOL, tests/test_gpu_fullsize.py); "
                         "not the 1e-4 parity path"}
            # configs[1] end to end: the SAME training iteration (G step with the bf16 forward + the bf16 backward sweep / weight
            # gradient, D and mask-D steps) on a fresh generator, so that the headline training leg below starts from untouched
            # weights; gradient parity of this mode: tests/test_gpu_backward.py (bf16 rows), tests/test_gpu_fullsize.py
            if args.train_steps > 0 and not args.no_disc:
                import copy
                ab = copy.copy(args)
                ab.train_steps = max(5, min(20, args.train_steps))
                g_b, d_b = build_models(R, S, I, K, "bf16", device)
                g_b.train()
                bf16_mode["training"] = bench_training(ab, g_b, d_b, device, 1, torch.cuda.synchronize, False, sub_legs="render")
                bf16_mode["training"]["what"] = ("Trainer.train_step with the generator in the bf16 operand mode: bf16 forward kernel, "
                                                 "mlp_bwd_sweep_kernel<bf16> (16-bit scratch slots: bf16 values, unorm16 phases) + "
                                                 "mlp_wgrad_f16_kernel<bf16 operands> backward; discriminators unchanged")
                del g_b, d_b
        except Exception as ex:
            bf16_mode = {**(bf16_mode or {}), "error": f"{type(ex).__name__}: {ex}"}
        finally:
            timer_on[0] = False
            gen.renderer.pack.set_precision(args.precision)
    timer.override = main_kernel_ms

    # headline reductions first: nothing after this point can take the timed result away
    t = torch.tensor(dts, device=device, dtype=torch.float64)
    de = torch.tensor([d_img_s_eager or 0.0], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(de, op=dist.ReduceOp.SUM)
    args._d_images_per_s_eager = float(de) if d_img_s_eager else None
    dd = torch.tensor([d_img_s or 0.0], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # per repeat: the slowest rank
        dist.all_reduce(dd, op=dist.ReduceOp.SUM)
    dts = [float(x) for x in t]
    dt = sum(dts) / repeats                       # mean duration of one K-step region
    rays_per_step = world * B * R * R
    value = rays_per_step * args.steps / dt
    spread = {"repeats": repeats, "timed_seconds": sum(dts),
              "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_median": float(np.median(dts)) / args.steps * 1e3,
              "ms_per_step_max": max(dts) / args.steps * 1e3}
    line = None

    # ---- full training iteration (3 renders, 6 D forwards, 3 backward + optimiser steps, flat-gradient
    #      all-reduce per network when N > 1); reported next to the headline, not part of `value`.
    #      A watchdog on every rank turns a stuck collective in this SECONDARY leg into "training: timed out"
    #      instead of a hung job without a bench line.
    train = None
    if args.train_steps > 0 and not args.no_disc:
        import threading

        line_ref = []
        if rank == 0:
            line_ref.append(build_line(args, value, dt, world, timer, float(dd) if d_img_s else None, None, distributed, bf16_mode, spread))

        def _watchdog():  # a thread, not SIGALRM: the main thread would be blocked inside a collective / synchronize
            if rank == 0 and line_ref:
                line_ref[0]["training"] = {"error": f"timed out after {args.train_timeout} s"}
                print(json.dumps(line_ref[0]), flush=True)
            os._exit(0)

        dog = threading.Timer(args.train_timeout, _watchdog)
        dog.daemon = True
        dog.start()
        try:
            train = bench_training(args, gen, disc, device, world, barrier, distributed)
        except Exception as ex:  # never lose the headline line because of the secondary measurement
            train = {"error": f"{type(ex).__name__}: {ex}"}
        dog.cancel()

    rank_devices = None
    if distributed:   # (rank, LOCAL_RANK, device it would take un-overridden, device it runs on): one small object gather
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, [rank, local_rank, local_rank, dev_index])
    if rank == 0:
        line = build_line(args, value, dt, world, timer, float(dd) if d_img_s else None, train, distributed, bf16_mode, spread)
        if rank_devices is not None:
            line["rank_devices"] = {"columns": ["rank", "local_rank", "cuda_index_by_local_rank", "cuda_index_used"], "rows": rank_devices}
        if world == 1 and not args.no_extras and not args.no_disc:
            try:
                line["extras"] = extras(args, gen, device)
            except Exception as ex:  # never lose the headline line because of a secondary measurement
                line["extras"] = {"error": f"{type(ex).__name__}: {ex}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.res, args.samples, args.importance, args.up_steps, args.batch)
        elif world > 1:  # rank-0-at-N=1 measurements: an explicit pointer instead of an absent key
            line["cpu_baseline"] = "see the N=1 line (the oracle is timed on rank 0 at N=1 only)"
            line["extras"] = "see the N=1 line (secondary single-GPU legs)"
            line["dist_backend"] = backend + (" (every rank on cuda:0: rehearsal of the N > 1 path on one GPU)" if os.environ.get("OI_BENCH_ONE_DEVICE") == "1" else "")
    if distributed:
        dist_barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio, which is
        # block-buffered on a pipe and would otherwise surface after it at exit
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


def _ev_time(fn, n, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n  # ms, GPU side (includes launch gaps when the host cannot keep up)


def extras(args, gen, device):
    """Secondary legs (N = 1 only; each a few hundred ms): driver-observed copies of what tools/ measures.
    c5_mlp          SURVEY 8d(iii): FiLM-SIREN MLP points/s at 2^21 points, sdf-only and full pass
    discriminator   SURVEY 8d(ii): ADADiscriminatorView forward at B in {1, 4, 64} (64^2) with its roofline: fp32-MFMA
                    FLOPs at B = 64, the 11.25 MB weight read at B = 1
    training_shipped_config  the reference's shipped training configuration (configs/train.yaml:70-76, 86, 102, 131:
                    128^2 crop, 16 + 4 samples, K = 1, discriminators at 128^2, batch 1 per GPU) -- the only configuration
                    the reference publishes a speed for (README.md:49: ~2.31 it/s on 2 x RTX 3090)
    inference       scripts/test.py -res 128 -depth 16 (256 + 64 samples per ray, multi-chunk), frames/s"""
    from oi_amd import ops
    from oi_amd.config import build_from_config
    out = {}
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    # ---- C5
    pack = gen.renderer.pack
    n = 1 << 21
    pts = (torch.rand(n, 3, device=device) * 2 - 1) * 0.9
    with torch.no_grad():
        _, gamma, beta = pack.film(z=torch.randn(1, 64, device=device))
        full = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, True, True, False, None)
        scratch = full[-1]
        ms_sdf = _ev_time(lambda: ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig), 10)
        ms_full = _ev_time(lambda: ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, True, True,
                                                   False, scratch), 10)
    peak = PEAK_TFLOPS[args.precision]
    out["c5_mlp"] = {"points": n, "precision": args.precision,
                     "sdf_only": {"ms": ms_sdf, "points_per_s": n / ms_sdf * 1e3, "algorithmic_TFLOP_per_s": n * F_SDF / ms_sdf / 1e9,
                                  "frac_of_mfma_peak": n * F_SDF / ms_sdf / 1e9 / peak},
                     "full": {"ms": ms_full, "points_per_s": n / ms_full * 1e3,
                              "algorithmic_TFLOP_per_s": n * (F_SDF + F_GRAD + F_COL) / ms_full / 1e9,
                              "frac_of_mfma_peak": n * (F_SDF + F_GRAD + F_COL) / ms_full / 1e9 / peak}}
    # C5 at the size BASELINE.json states: 2^20 rays x 512 samples = 2^29 points, sdf-only (123.7 TFLOP), as
    # 256 back-to-back launches over the resident 2^21-point buffer (the full point set would be 6 GB of inputs for nothing
    # the kernel could tell apart); one pass, HIP events around all of it
    with torch.no_grad():
        ms_c5 = _ev_time(lambda: [ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig)
                                  for _ in range(256)], 1, warm=0)
    n5 = n * 256
    out["c5_mlp"]["full_size_sdf_only"] = {"points": n5, "launches": 256, "ms": ms_c5, "points_per_s": n5 / ms_c5 * 1e3,
                                           "algorithmic_TFLOP_per_s": n5 * F_SDF / ms_c5 / 1e9,
                                           "frac_of_mfma_peak": n5 * F_SDF / ms_c5 / 1e9 / peak}
    del pts, full, scratch
    # ---- discriminator batch sweep (forward, eval, no grad)
    disc = build_from_config(net("src.models.discriminator.ADADiscriminatorView",
                                 aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=64,
                                 in_dim=3, last_bias=False, n_feat=512, out_dim=7, out_dim_latent=0, out_dim_position=6)).to(device).eval()
    d_flops, d_wbytes = 207.7e6, 11.25e6   # per image forward, fp32 weight bytes
    sweep = {}
    for bsz in (1, 4, 64):
        x = torch.rand(bsz, 3, 64, 64, device=device)
        with torch.no_grad():
            ms = _ev_time(lambda: disc(x, it=0), 30 if bsz < 64 else 10)
        sweep[f"B{bsz}"] = {"ms": ms, "images_per_s": bsz / ms * 1e3,
                            "roofline": {"mfma_fp32": {"achieved_TFLOP_per_s": bsz * d_flops / ms / 1e9, "peak": 157.3,
                                                       "frac": bsz * d_flops / ms / 1e9 / 157.3},
                                         # the large-batch convolutions run on the fp16 matrix cores (3 MFMAs per MAC)
                                         "mfma_f16": {"achieved_TFLOP_per_s": bsz * d_flops / ms / 1e9, "peak": 2500.0,
                                                      "frac": bsz * d_flops / ms / 1e9 / 2500.0,
                                                      "executed_mfma_frac_of_peak": 3 * bsz * d_flops / ms / 1e9 / 2500.0},
                                         "weight_read_hbm": {"achieved_GB_per_s": d_wbytes / ms / 1e6, "peak": 8000.0,
                                                             "frac": d_wbytes / ms / 1e6 / 8000.0}}}
    # the shipped network (configs/train.yaml:78-102: 128^2, five blocks 3 -> 32 -> ... -> 512 -> 7) at batch 1: csrc/disc_small.hip since
    # round 6 (oi_disc_fwd_small128), against the general layer-by-layer chain it took until then
    import oi_amd.discriminator as _DM
    disc128 = build_from_config(net("src.models.discriminator.ADADiscriminatorView",
                                    aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=128,
                                    in_dim=3, last_bias=False, n_feat=512, out_dim=7, out_dim_latent=0, out_dim_position=6)).to(device).eval()
    x128 = torch.rand(1, 3, 128, 128, device=device)
    with torch.no_grad():
        ms128 = _ev_time(lambda: disc128(x128, it=0), 30)
        _DM.SMALL_PATH_128 = False
        _DM._FAST_ADA.pop(disc128, None)
        try:
            ms128_general = _ev_time(lambda: disc128(x128, it=0), 30)
        finally:
            _DM.SMALL_PATH_128 = True
    sweep["B1_128"] = {"ms": ms128, "images_per_s": 1e3 / ms128, "general_chain_ms": ms128_general,
                       "what": "the SHIPPED 128^2 / five-block network at batch 1 (281.1 MFLOP per image): five launches of csrc/disc_small.hip; "
                               "general_chain_ms = the same forward through the layer-by-layer chain of csrc/disc.hip",
                       "roofline": {"mfma_fp32": {"achieved_TFLOP_per_s": 281.1e6 / ms128 / 1e9, "peak": 157.3, "frac": 281.1e6 / ms128 / 1e9 / 157.3}}}
    del disc128, x128
    out["discriminator"] = {"what": "ADADiscriminatorView forward (ADA xint + scale, 5 conv4x4 s2 + head), 64^2; batch <= 4: csrc/disc_small.hip "
                                    "(four launches, fp32 VALU, weight-stream bound); batch >= 16: csrc/disc_large.hip (NHWC fp16 limb planes, "
                                    "packed weight images, f16x3 = 22-bit operands / fp32 accumulate on the matrix cores, fixed-order "
                                    "split-K); wall time per forward incl. host launches; bound: "
                                    "weight read at B = 1; at B = 64 both yardsticks: the fp32-MFMA peak (what an fp32 convolution could reach) and the fp16-MFMA "
                                    "peak of the unit the tiled kernel actually runs on", **sweep}
    del disc
    # ---- shipped training configuration
    import copy
    a2 = copy.copy(args)
    a2.res, a2.samples, a2.importance, a2.up_steps, a2.batch = 128, 16, 4, 1, 1
    a2.train_steps = max(5, min(20, args.train_steps or 10))
    g2, d2 = build_models(128, 16, 4, 1, args.precision, device)
    g2.train()
    tr = bench_training(a2, g2, d2, device, 1, torch.cuda.synchronize, False, sub_legs=False)
    out["training_shipped_config"] = {
        "config": "configs/train.yaml: 128x128 crop, 16 + 4 samples/ray, 1 up-sampling step, batch 1 per GPU, RGB-D + mask "
                  "discriminators at 128^2 (n_feat 512), Adam 2e-5 / RMSprop 1e-4",
        "it_per_s": tr["it_per_s"], "ms_per_it": tr["ms_per_it"], "steps": tr["steps"],
        "published": {"it_per_s": 2.31, "hardware": "2 x GeForce RTX 3090, DDP x2 (README.md:49: 100k iterations in ~12 h)"},
        "vs_baseline": tr["it_per_s"] / 2.31,
        "note": "per-GPU iteration rate on ONE MI355X (batch 1) against the reference's two-GPU DDP run (batch 1 per GPU, same "
                "per-GPU work, all-reduce included there): different hardware, context only"}
    del g2, d2
    # ---- inference driver leg
    from oi_amd import inference
    g3, _ = build_models(128, 256, 64, 1, args.precision, device)
    g3.eval()
    zs = [torch.randn(64) for _ in range(4)]
    np.random.seed(0)
    b2w = torch.tensor(g3.pose_prior(1), dtype=torch.float32)[0]
    inference.render_frames(g3, zs[:1], [b2w], keys=("image",))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inference.render_frames(g3, zs, [b2w] * 4, keys=("image", "normal_map", "shading_map"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    out["inference"] = {"what": "scripts/test.py -res 128 -depth 16: 128x128 frame, 256 + 64 samples per ray, eval, ray chunks "
                                "of MAX_RAY_BATCH_SIZE / bs (generator.py:286-305)", "frames": 4, "s_per_frame": dt,
                        "rays_per_s": 128 * 128 / dt, "points_per_frame": 128 * 128 * 320}
    return out


def csrc_digest():
    import importlib.util
    spec = importlib.util.spec_from_file_location("oi_build", os.path.join(ROOT, "object-intrinsics_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._digest()


TRAFFIC_FILE = "r*_traffic.json"   # the newest round's file whose digest matches the built sources


def measured_traffic(args, precision=None):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r3_traffic.json,
    written by tools/traffic_json.py: FETCH_SIZE x2 -- the gfx950 correction of MI355X_MICROARCH.md -- + WRITE_SIZE).
    Counters cannot be collected from inside the timed process, so the figure is tied to the kernel sources by digest:
    a build whose csrc/ differs from the profiled one reports traffic = null instead of a stale number."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", TRAFFIC_FILE)),
                   key=lambda p_: int(re.search(r"r(\d+)_traffic", p_).group(1)), reverse=True)
    if not paths:
        return None, f"profiles/{TRAFFIC_FILE} missing"
    key = f"{precision or args.precision}:{args.batch}x{args.res}x{args.res}:{args.samples}+{args.importance}"
    dig, why = csrc_digest(), None
    for path in paths:   # newest round first
        rec = json.load(open(path))
        ent = rec.get("entries", {}).get(key)
        name = os.path.basename(path)
        if ent is None:
            why = why or f"no PMC record for {key} in profiles/{name}"
        elif rec.get("csrc_digest") != dig:
            why = why or f"csrc/ changed since the PMC passes of profiles/{name} were taken (re-run tools/refresh_profiles.sh)"
        else:
            return float(ent["bytes_per_launch"]), ent["source"]
    return None, why


def build_line(args, value, dt, world, timer, d_img_s, train, distributed, bf16_mode=None, spread=None):
    R, S, I, K, B = args.res, args.samples, args.importance, args.up_steps, args.batch
    if True:
        n_pts = B * R * R * (S + I)
        kern_ms = timer.mean_ms()
        flops = n_pts * (F_SDF + F_GRAD + F_COL)
        achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms else None
        peak = PEAK_TFLOPS[args.precision]
        traffic, traffic_src = measured_traffic(args)
        line = {
            "metric": "rendered rays/sec (64x64 img, 128 samples/ray) + D-images/sec",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "timing": spread,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3 (fp32 split into 2 bf16 MFMA operands, fp32 accumulate)",
                      "bf16x6": "f32 via bf16x6 (fp32 operands split 3-way, 6 bf16 MFMAs per product, fp32 accumulate: "
                                "fp32-exact contractions, 1e-4 parity path)",
                      "f16x3": "f32 via f16x3 (fp32 operands split into 2 fp16 limbs = 22 mantissa bits, 3 fp16 MFMAs "
                               "per product, fp32 accumulate: 2^-22 relative per product, 1e-4 parity path)",
                      "bf16": "bf16"}[args.precision],
            "data": "synthetic (random poses/latents/backgrounds from the data/example prior; sphere-initialised SDF "
                    "weights, seeded default-init colour/discriminator weights)",
            "config": {"workload": f"{ {(64, 64, 64, 1): 'C2', (128, 128, 128, 4): 'C4'}.get((R, S, I, K), 'custom') }: {B}x{R}x{R} crop per GPU, {S}+{I} samples/ray, {K} up-sampling step(s), "
                                   f"Generator.forward (render + Phong maps) + ADADiscriminatorView forward",
                       "rays_per_step_per_gpu": B * R * R, "points_per_step_per_gpu": n_pts,
                       "parallelism": f"dp{world} (independent renders, no data-path collective)"},
            "d_images_per_s": d_img_s,
            "d_images_per_s_what": "ADADiscriminatorView forward, batch 1 per GPU, through oi_amd.graphed.GraphedDForward: at batch "
                                   "<= 4 a plan held by the library (oi_disc_graph_*: four launches per image, image pointer "
                                   "passed per call; OI_DISC_LAUNCH=graph replays them as one hipGraph); d_images_per_s_eager = the "
                                   "module's own forward (the reference's call), call by call.  Since round 6 both draw the "
                                   "augmentation parameters INSIDE the library from one seed of numpy's stream per call "
                                   "(oi_disc_graph_launch_ada): no numpy / Python arithmetic per image",
            "d_images_per_s_eager": getattr(args, "_d_images_per_s_eager", None),
            "training": train,
            "bf16_mode": bf16_mode,
            "roofline": {"bound": "mfma",
                         "kernel": ("sdf_mlp_full3_kernel (register-resident: sdf + d sdf/dx + albedo at the fine samples; since round 6 "
                                    "its per-element table blob is formed by the step's prep launch, so the HIP-event pair covers "
                                    "this kernel alone -- with OI_STEP_TAIL=0 it also covers film_blob_f3_kernel, ~6 us)"
                                    if args.precision == "f16x3" else "sdf_mlp_kernel<full> (sdf + d sdf/dx + albedo at the fine samples)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_flops_per_launch": flops, "kernel_ms": kern_ms,
                         "kernel_ms_samples": getattr(args, "_kernel_samples", None),
                         "algorithmic_bytes_per_launch": n_pts * 40,  # 12 B point in, 28 B sdf + gradient + albedo out
                         # the same launch against the HBM roofline (8 TB/s, MI355X_MICROARCH.md): PMC traffic / live duration
                         "hbm": ({"achieved": traffic / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": traffic / (kern_ms * 1e-3) / 8e12} if (kern_ms and traffic) else None),
                         "executed_mfma_frac_of_peak": (achieved * MFMA_PER_MAC[args.precision] / peak) if achieved else None,
                         "vs_native_fp32_mfma_peak": (achieved / 157.3) if achieved else None,
                         "note": "kernel_ms = mean of HIP-event pairs around every 4th launch of this kernel inside the timed region "
                                 "(kernel_ms_samples pairs; a pair costs its launch ~11 us of idle stream); "
                                 "algorithmic = GEMM MACs x2 of sdf fwd + analytic gradient sweep + colour head per "
                                 "point. peak = dense MFMA peak of the unit the mode runs on (fp32 MFMA 157.3, fp16 / bf16 MFMA "
                                 "2500 TFLOP/s); f16x3 / bf16x3 / bf16x6 execute 3 / 3 / 6 16-bit MFMAs per algorithmic MAC "
                                 "(executed_mfma_frac_of_peak), so their frac is bounded by 1/3 / 1/3 / 1/6; "
                                 "vs_native_fp32_mfma_peak compares the fp32-exact result rate with the native fp32 roofline"},
        }
    return line


if __name__ == "__main__":
    main()
