"""Per-phase shader-clock profile of the backward sweep (library built with -DOI_BWD_PROF:
tools/dbg/build_variants.sh mlp_bwd.hip bprof "-DOI_BWD_PROF", run with OI_LIB=.../liboi_bprof.so)."""
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
names = ["prologue + colour head", "up: layer 0 (VALU)", "up: dma_sync", "up: stage issue + vbar epilogue/stores",
         "up: the two products", "up: phi epilogue + stores", "up->down turn (w_sigma, g_8, first loads)", "down: dma_sync",
         "down: epilogue (sincos, reductions, stores)", "down: issue next loads", "down: the two products + copies",
         "down: barrier + row flush"]
nw = buf[13]
print(f"waves {nw}, mean ticks per wave {buf[12] / nw:.0f}")
for i, nm in enumerate(names):
    v = buf[i] + (sum(buf[14:19]) if i == 0 else 0)
    print(f"  {nm:55s} {v / nw:10.0f}  {100 * v / buf[12]:5.1f} %")
sub = ["colour: first loads + tables + dma_sync", "colour: product 1 + barrier + next DMA issue", "colour: epilogue (trig, 16 stores, 24 row sums)",
       "colour: flushes (atomics, two barriers)", "colour: product 2", "colour: a_8 term to registers / slot"]
print("of the prologue + colour head:")
for i, nm in enumerate(sub):
    v = buf[14 + i] if i < 5 else buf[0]   # (the marks share one running clock: slot 0 is what follows the last sub-mark)
    print(f"    {nm:53s} {v / nw:10.0f}  {100 * v / buf[12]:5.1f} %")
span = buf[22] - ((1 << 62) - buf[21])
print(f"100 MHz clock: workgroup lifetimes / 256 CUs = {buf[20] / 256 / 100:.1f} us against a span of {span / 100:.1f} us over the launches read "
      f"(first start to last end; ONE launch only when the readback follows a reset + one backward)")
# per-CU timelines: gaps between consecutive workgroups of a CU, and when each CU ran dry
wg = (ctypes.c_ulonglong * (4 * 4096))()
raw.oi_prof_bwd_read_wg(wg)
import collections
ntile = (n + 255) // 256 * B
per_cu = collections.defaultdict(list)
for i in range(ntile):
    st, en, hw, xcc = wg[4 * i], wg[4 * i + 1], wg[4 * i + 2], wg[4 * i + 3]
    per_cu[(xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append((st, en))
t0 = min(v[0][0] for v in map(sorted, per_cu.values()))
gaps, ends, counts, lifes = [], [], [], []
for k, v in per_cu.items():
    v.sort()
    counts.append(len(v))
    ends.append(v[-1][1] - t0)
    lifes += [b - a for a, b in v]
    gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
import statistics as st_
print(f"CUs seen {len(per_cu)}, workgroups per CU min / max {min(counts)} / {max(counts)}; lifetime mean {st_.mean(lifes) / 100:.1f} us "
      f"(min {min(lifes) / 100:.1f}, max {max(lifes) / 100:.1f})")
print(f"gap between consecutive workgroups of a CU: mean {st_.mean(gaps) / 100:.2f} us, median {st_.median(gaps) / 100:.2f}, max {max(gaps) / 100:.2f}")
ends.sort()
print(f"a CU's last workgroup ends at: min {ends[0] / 100:.1f} us, median {ends[len(ends) // 2] / 100:.1f}, max {ends[-1] / 100:.1f} "
      f"(mean idle tail {(ends[-1] - st_.mean(ends)) / 100:.1f} us)")
