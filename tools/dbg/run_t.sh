#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_c5.py --modes f16x3 2>&1 | grep f16x3
timeout 300 python tools/dbg/time_bwd_k.py 2>&1 | tail -1
} > gpurun_out/t.log 2>&1 < /dev/null
