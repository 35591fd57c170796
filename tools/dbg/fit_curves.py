"""Loss curves of the generator-only fit of tests/test_gpu_backward.py::_generator_fit under optimiser variants (run-to-run noise)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
import test_gpu_backward as T
from oi_amd.optim import FusedAdam
import oi_amd.optim as OPT

def fit(prec, betas, lr, steps=200, decay_at=None):
    orig = OPT.FusedAdam
    class A(orig):
        def __init__(self, params, lr=1e-4, betas=(0.0, 0.9)):
            super().__init__(params, lr=LR[0], betas=BETAS[0])
    LR[0], BETAS[0] = lr, betas
    T_FusedAdam = A
    import oi_amd.optim
    oi_amd.optim.FusedAdam = A
    try:
        return T._generator_fit(prec, steps)
    finally:
        oi_amd.optim.FusedAdam = orig
LR, BETAS = [1e-4], [(0.0, 0.9)]
for betas, lr in (((0.0, 0.9), 1e-4), ((0.9, 0.999), 1e-4), ((0.9, 0.999), 3e-5), ((0.0, 0.9), 2e-5)):
    for prec in ("f32", "bf16"):
        for rep in range(3):
            l = fit(prec, betas, lr)
            print(f"betas {betas} lr {lr:g} {prec} rep {rep}: first {l[0]:.4f} " + " ".join(f"{l[i:i+20].mean():.4f}" for i in range(20, 200, 20)) + f"  last50/first {l[-50:].mean() / l[0]:.3f} min {l.min() / l[0]:.3f}")
