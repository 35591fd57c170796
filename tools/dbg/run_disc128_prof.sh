#!/bin/bash
# kernel times of the batch-1 ADADiscriminatorView forward at 128 x 128 (small path vs the general chain) + the shipped training configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_d128; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_d128 -- python $R/tools/bench_disc_eager.py --res 128 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_d128 $O/disc128_kernel_stats.txt > /dev/null; head -16 $O/disc128_kernel_stats.txt | cut -c1-130
cd $R
python - <<'PY'
import sys, time, os
sys.path.insert(0, "object-intrinsics_amd")
import torch
import oi_amd.discriminator as DM
from oi_amd.config import build_from_config
net = lambda t, **kw: {"__target__": t, "kwargs": kw}
disc = build_from_config(net("src.models.discriminator.ADADiscriminatorView", out_dim_latent=0, out_dim_position=6,
       aug=net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1), aug_p=1, img_size=128, in_dim=3, last_bias=False, n_feat=512, out_dim=7)).cuda().eval()
x = torch.rand(1, 3, 128, 128, device="cuda")
def rate(n=2000):
    with torch.no_grad():
        for _ in range(50): disc(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): disc(x)
        torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)
for on in (True, False, True, False):
    DM.SMALL_PATH_128 = on; DM._FAST_ADA.pop(disc, None)
    print(f"128 x 128 batch 1, small path {on}: {rate():8.0f} images/s")
PY
for v in 1 0; do
OI_SMALL128=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --min-seconds 0.01 --train-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['extras']['training_shipped_config']; print('small128=$v shipped config', 'ms_per_it', round(t['ms_per_it'],3), 'it_per_s', round(t['it_per_s'],1))"
done
