#!/bin/bash
# same-box A/B of the bf16 forward kernel: per-element images (default) vs OI_BF16_PRESCALE=0 (round-4 kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py -m gpu -x -q -k "bf16 or golden or ragged" 2>&1 | tail -15) > $O/b3p_tests.log
for v in 1 0 1 0; do
  echo "prescale=$v" >> $O/b3p_c5.log
  OI_BF16_PRESCALE=$v python tools/bench_c5.py --modes bf16 >> $O/b3p_c5.log 2>&1
done
for v in 1 0; do
  OI_BF16_PRESCALE=$v python bench.py --precision bf16 --no-cpu-baseline --no-bf16 --no-extras --train-steps 20 2>/dev/null | tail -1 > $O/b3p_bench_$v.json
done
cat $O/b3p_tests.log; cat $O/b3p_c5.log
python - <<'P'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
for v in (1,0):
    d=json.load(open(f'{O}/b3p_bench_{v}.json'))
    print(v, d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('frac'), d.get('training',{}).get('ms_per_it'))
P
