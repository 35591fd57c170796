#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python bench.py --steps 5 --warmup 2 --min-seconds 0.2 --no-cpu-baseline --no-bf16 --no-extras --train-steps 600 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['training']
print('600 iterations:', t['it_per_s'], t['ms_per_it'], t['finite'])
"
timeout 900 python bench.py --res 128 --samples 16 --importance 4 --up-steps 1 --steps 5 --warmup 2 --min-seconds 0.2 --no-cpu-baseline --no-bf16 --no-extras --train-steps 300 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['training']
print('shipped config 300 iterations:', t['it_per_s'], t['ms_per_it'], t['finite'])
"
} > gpurun_out/long.log 2>&1 < /dev/null
