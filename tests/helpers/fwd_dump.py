#!/usr/bin/env python3
"""Outputs of one seeded full forward (sdf, d sdf/dx, albedo, features) and one sdf-only forward of the library selected by
OI_LIB, saved for a bit-for-bit comparison between library variants:  fwd_dump.py out.pt [mode]  |  fwd_dump.py --cmp a.pt b.pt
Test infrastructure: tests/test_gpu_kernels.py::test_persistent_workgroups_bit_identical runs it in two subprocesses (the
launch mode is read once per process); tools/dbg/run_fwd_ab.sh and run_v2_persist_ab.sh use it for same-box A/Bs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/helpers/ -> repo root
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch

if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k].view(torch.int32), b[k].view(torch.int32))
        d = (a[k].double() - b[k].double()).abs().max().item()
        print(f"{k:10s} {'bit-identical' if same else 'DIFFERENT'}  max |diff| {d:.3e}  max |a| {a[k].abs().max().item():.3e}")
    sys.exit(0)

from conftest import load_golden
from oi_amd import ops
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack

kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
sdf = ShapeNetwork(os.path.join(ROOT, "tests", "golden", "weights_sdf.npz"), **kw).cuda()
col = ColorNetwork(**kw); col.load_state_dict(load_golden("weights_color")); col = col.cuda()
g = torch.Generator(device="cuda").manual_seed(7)
B, n = 2, 100_003
pts = (torch.rand(B * n, 3, device="cuda", generator=g) * 2 - 1) * 0.9
z = torch.randn(B, 64, device="cuda", generator=g)
mode, _, trig = (sys.argv[2] if len(sys.argv) > 2 else "f16x3").partition(":")
pack = FieldPack(sdf, col, mode)
if trig:
    pack.set_precision(mode, fast_trig=(trig == "fast"))
with torch.no_grad():
    _, gamma, beta = pack.film(z=z)
    full = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig, True, True, True, None)
    only = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, B, pack.prec, pack.fast_trig, False, False, False, None)
out = {f"full{i}": t.float().cpu() for i, t in enumerate(full[:-1]) if torch.is_tensor(t)}
out.update({f"sdf{i}": t.float().cpu() for i, t in enumerate(only) if torch.is_tensor(t) and t.numel() == B * n})
torch.save(out, sys.argv[1])
print({k: tuple(v.shape) for k, v in out.items()})
