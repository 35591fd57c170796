#!/usr/bin/env python3
"""Print the worst-case error of every renderer output key vs the reference's golden vectors (tests/golden/f4_render)
for each MFMA operand mode: the margin each mode has against the 1e-4 parity bar.  Needs a GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from test_gpu_modules import make_renderer, KEYS

g = load_golden("f4_render")
col_sd = load_golden("weights_color")
for prec in sys.argv[1:] or ["f32", "bf16x6", "f16x3", "bf16x3"]:
    prec, _, trig = prec.partition(":")  # "f16x3:fast" forces the unreduced hardware sin/cos
    r = make_renderer(col_sd, 16, 16, 1, prec)
    if trig:
        r.pack.set_precision(prec, fast_trig=(trig == "fast"))
        prec = f"{prec}:{trig}"
    worst = {}
    for tag, car in (("c0p0", 0.0), ("c0p5", 0.5), ("c1p0", 1.0)):
        with torch.no_grad():
            out = r.render(g["rays_o"].cuda(), g["rays_d"].cuda(), g["near"].cuda(), g["far"].cuda(), perturb_overwrite=0,
                           cos_anneal_ratio=car, z=None, w=g["w"].cuda())
        for k in KEYS:
            ref = g[f"{tag}_{k}"]
            scale = max(1.0, float(ref.abs().max())) if k == "gradients" else 1.0
            worst[k] = max(worst.get(k, 0.0), float((out[k].cpu() - ref).abs().max()) / scale)
    print(f"{prec:10s} max {max(worst.values()):.2e}  " + "  ".join(f"{k}={v:.1e}" for k, v in worst.items()))
