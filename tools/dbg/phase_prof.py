"""Per-phase shader-clock profile of the forward MLP kernel (needs a library built with OI_FLAGS=-DOI_PROF)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
from oi_amd import lib
gen, disc = bench.build_models(64, 64, 64, 1, sys.argv[1] if len(sys.argv) > 1 else "f16x3", torch.device("cuda"))
gen.train()
L = lib.load()
raw = ctypes.CDLL(lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for i in range(3):
        gen(bs=1, it=i, data={})
    raw.oi_prof_read(buf, 1)
    for i in range(5):
        gen(bs=1, it=i, data={})
    raw.oi_prof_read(buf, 0)
names = ["layer0 VALU", "fwd GEMM", "fwd stage_late (barrier+DMA issue)", "fwd FiLM/sin + stores", "fwd ring_sync (vmcnt0+barrier)",
         "rev c-load+mul+normalise", "rev GEMM", "rev stage_late", "rev copy + ring_sync", "tail (sdf, layer-0 grad, colour)"]
n = buf[11]
tot = buf[10] / n
print(f"waves {n}, mean cycles per wave {tot:.0f} (shader clock 100 MHz ticks x?)")
for i, nm in enumerate(names):
    print(f"  {nm:40s} {buf[i] / n:10.0f}  {100 * buf[i] / buf[10]:5.1f} %")
