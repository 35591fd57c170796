"""Per-phase shader-clock profile of the weight-gradient GEMM (library built with -DOI_WG_PROF:
tools/dbg/build_variants.sh mlp_bwd.hip wgprof "-DOI_WG_PROF", run with OI_LIB=.../liboi_wgprof.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from test_gpu_backward import NET_KW, SDF_NPZ
from oi_amd import lib
from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
from oi_amd.autograd import sdf_mlp
col_sd = load_golden("weights_color")
sdf_net = ShapeNetwork(SDF_NPZ, **NET_KW).cuda(); col_net = ColorNetwork(**NET_KW); col_net.load_state_dict(col_sd); col_net = col_net.cuda()
pack = FieldPack(sdf_net, col_net, "f16x3")
B, n = 1, 524288
pts = (torch.rand(B * n, 3, device="cuda") * 2 - 1)
w = torch.randn(B, 64, device="cuda").requires_grad_(True)
raw = ctypes.CDLL(lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 24)()
for it in range(4):
    _, gamma, beta = pack.film(w=w)
    sdf, grad, rgb, _ = sdf_mlp(pack, pts, gamma, beta, B, True, True, False)
    loss = sdf.sum() + grad.sum() + rgb.sum()
    raw.oi_prof_bwd_read(buf, 1)
    loss.backward()
raw.oi_prof_bwd_read(buf, 0)
names = ["wait for the tile's loads + sin / cos", "barrier 1", "LDS writes pair 0", "barrier 2 + fragments pair 0",
         "barrier 3 + LDS writes pair 1 + next tile's requests", "MFMAs pair 0", "barrier 4 + fragments pair 1",
         "barrier 5 + MFMAs pair 1"]
if os.environ.get("OI_WG_TR", "1") != "0":   # the transposing-read build (round 6): five phases
    names = ["barrier A (the previous tile's reads; the MFMAs behind them)", "wait for the loads + unpack, sin / cos, split, 32 ds_write_b64",
             "barrier B", "the tile two ahead: 19 buffer loads issued", "80 transposing reads + 48 MFMAs", "-", "-", "-"]
nt = buf[13]
print(f"wave-tiles {nt}, mean ticks per tile (layer matrices) {buf[12] / nt:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:55s} {buf[i] / nt:10.0f}  {100 * buf[i] / buf[12]:5.1f} %")
