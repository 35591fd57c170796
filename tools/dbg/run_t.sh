#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "graphed_discriminator_forward" 2>&1 | tail -15
timeout 600 python bench.py --steps 10 --warmup 3 --min-seconds 0.3 --no-cpu-baseline --no-bf16 --no-extras --train-steps 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['d_images_per_s'], d['d_images_per_s_eager'])
"
} > gpurun_out/t.log 2>&1 < /dev/null
