#!/usr/bin/env python3
"""Measured gradient margins (needs a GPU): runs the gradient parity tests with OI_MARGIN_OUT set and prints, per case
and operand mode, the worst relative error over the checked tensors, the tensor it occurs in, and the median -- the
table of DESIGN.md section 5.  The test tolerances are derived from the native-fp32 column (<= 3x).

    python tools/grad_margin.py [out.json]
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/oi_margins.json"
env = dict(os.environ, OI_MARGIN_OUT=out)
sel = "full_size_render_properties or mlp_backward or f6 or f7 or f9 or f14 or c2_size or c4_size or composite_backward or bf16 or color_network or reference_style or ada_geom_separable"
r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu", "-k", sel, "-p",
                    "no:cacheprovider"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout.strip().splitlines()[-1])
m = json.load(open(out))
print(f"{'case':58s} {'tensors':>7s} {'worst':>9s} {'median':>9s}  worst tensor")
for case in sorted(m):
    v = m[case]
    k = max(v, key=v.get)
    print(f"{case:58s} {len(v):7d} {v[k]:9.2e} {statistics.median(v.values()):9.2e}  {k}")
