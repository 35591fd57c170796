#!/bin/bash
mkdir -p gpurun_out
{
timeout 1200 python bench.py --res 128 --samples 128 --importance 128 --up-steps 4 --steps 5 --warmup 2 --min-seconds 0.2 --train-steps 4 --no-bf16 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['training']
print('C4:', d['value'], d['ms_per_step'], 'training', t['it_per_s'], t['ms_per_it'], t['render_fwd_bwd']['ms'], t['finite'])
"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -2
} > gpurun_out/c4.log 2>&1 < /dev/null
