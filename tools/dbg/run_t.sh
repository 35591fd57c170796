#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_ddp.py -m gpu -x -q -k "graphed or train or ddp or iteration" 2>&1 | tail -3
for v in 0 1 0 1; do echo "== side stream $v"; OI_D_SIDE_STREAM=$v timeout 600 python bench.py --steps 5 --warmup 2 --min-seconds 0.2 --no-cpu-baseline --no-bf16 --no-extras --train-steps 40 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['training']
print(t['it_per_s'], t['ms_per_it'], t['d_step']['ms'], t['finite'])
"; done
} > gpurun_out/t.log 2>&1 < /dev/null
