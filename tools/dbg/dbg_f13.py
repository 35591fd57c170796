import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os
sys.path[:0] = [ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests", ROOT + "/oracle"]
import pytest, torch
import oi_amd.discriminator as DM, oi_amd.generator as G, oi_amd.ops as ops
mode = MODE
if mode == "nosmall": DM.SMALL_PATH = False
if mode == "noprep": G.PREP_MAX_B = 0
if mode == "nostats": ops.FUSED_STATS = False
if mode == "noplanar":
    orig = ops.composite_fwd
    def cf(*a, image_planar=False, **k): return orig(*a, image_planar=False, **k)
    ops.composite_fwd = cf
sys.exit(pytest.main(["-q", "-x", "-m", "gpu", ROOT + "/tests/test_gpu_trainer_f13.py", "-k", "False"]))
'''
for mode in ("all", "nosmall", "noprep", "nostats", "noplanar"):
    src = code.replace("ROOT", repr(ROOT)).replace("MODE", repr(mode))
    p = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True)
    tail = [l for l in p.stdout.splitlines() if "AssertionError" in l or "passed" in l or "failed" in l]
    print(mode, "->", tail[-2:])
