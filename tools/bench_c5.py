#!/usr/bin/env python3
"""C5 micro-benchmark: FiLM-SIREN MLP points/s and algorithmic TFLOP/s on one GPU, for the
sdf-only pass and the full pass (sdf + d sdf/dx + albedo), every operand mode.  HIP-event timing around the launches."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from oi_amd import ops
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack

F_SDF, F_GRAD, F_COL = 230400, 230400, 34304
ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1 << 21, help="total points (BASELINE configs[4]: 2^20 rays x 512 = 2^29)")
ap.add_argument("--chunk", type=int, default=1 << 24, help="points per launch (bounds the 4.6 KB/point scratch of the full pass)")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--modes", default="f16x3,bf16x6,f32,bf16x3,bf16", help="comma list; append :poly / :fast to force the sincos flavour")
args = ap.parse_args()
kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
sdf = ShapeNetwork(os.path.join(ROOT, "tests", "golden", "weights_sdf.npz"), **kw).cuda()
col = ColorNetwork(**kw); col.load_state_dict(load_golden("weights_color")); col = col.cuda()
n = min(args.points, args.chunk)
launches = max(1, args.points // n)
pts = (torch.rand(n, 3, device="cuda") * 2 - 1) * 0.9
for mode in args.modes.split(","):
    mode, _, trig = mode.partition(":")
    pack = FieldPack(sdf, col, mode)
    if trig:
        pack.set_precision(mode, fast_trig=(trig == "fast"))
        mode = f"{mode}:{trig}"
    with torch.no_grad():
        _, gamma, beta = pack.film(z=torch.randn(1, 64, device="cuda"))
        scratch = None
        res = {}
        for full in (False, True):
            # warm-up by TIME (>= 0.3 s of launches): the first tens of milliseconds of a process run at ramping clocks, and
            # a case measured there reads up to 15 % slow (seen as a spurious "fast trig" gain of the second mode in the list)
            w0 = torch.cuda.Event(enable_timing=True); w0.record()
            while True:
                for _ in range(8):
                    out = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, full, full, False, scratch)
                    scratch = out[-1] if full else scratch
                w1 = torch.cuda.Event(enable_timing=True); w1.record(); w1.synchronize()
                if w0.elapsed_time(w1) > 300.0:
                    break
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters * launches):
                ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, full, full, False, scratch)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tot = n * launches
            fl = tot * (F_SDF + (F_GRAD + F_COL if full else 0))
            res["full" if full else "sdf_only"] = {"ms": round(ms, 4), "Mpoints_per_s": round(tot / ms / 1e3, 1),
                                                   "algorithmic_TFLOP_per_s": round(fl / ms / 1e9, 1)}
    print(json.dumps({"mode": mode, "points": n * launches, "points_per_launch": n, **res}))
