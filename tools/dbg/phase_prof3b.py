"""Per-phase shader-clock profile of the register-resident forward kernel (library built with -DOI_B3_PROF:
tools/dbg/build_variants.sh mlp_fwd3b.hip prof "-DOI_B3_PROF", run with OI_LIB=.../liboi_prof.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from oi_amd import ops, lib
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
sdf = ShapeNetwork(os.path.join(ROOT, "tests", "golden", "weights_sdf.npz"), **kw).cuda()
col = ColorNetwork(**kw); col.load_state_dict(load_golden("weights_color")); col = col.cuda()
n = 1 << 21
pts = (torch.rand(n, 3, device="cuda") * 2 - 1) * 0.9
pack = FieldPack(sdf, col, "bf16")
raw = ctypes.CDLL(lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    _, gamma, beta = pack.film(z=torch.randn(1, 64, device="cuda"))
    out = ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, True, True, False, None)
    raw.oi_prof3b_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.sdf_mlp_fwd(pts, pack.packed(), gamma, beta, 1, pack.prec, pack.fast_trig, True, True, False, out[-1])
    e1.record()
    raw.oi_prof3b_read(buf, 0)
    print(f"wall per launch {e0.elapsed_time(e1) / 5:.3f} ms (profiled build)")
names = ["layer 0 (VALU)", "layer products (MFMA windows + overlapped epilogues)", "ring_sync (vmcnt0 + barrier)",
         "exposed block-3 epilogues", "prologue (tables, FiLM rows, maxima; before the tile clock)", "heads / gradient / rest"]
nw = buf[7]
print(f"waves {nw}, mean ticks per wave {buf[6] / nw:.0f}")
if buf[9]:
    print(f"shader clock inside the kernel: {buf[8] / buf[9] * 100:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick; "
          f"mean wave lifetime {buf[9] / nw / 100:.1f} us)")
for i, nm in enumerate(names):
    print(f"  {nm:55s} {buf[i] / nw:10.0f}  {100 * buf[i] / buf[6]:5.1f} %")
