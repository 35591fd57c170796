#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/bench_c5.py 2>&1 | tail -12
OI_FWD_V2=1 timeout 300 python tools/bench_c5.py 2>&1 | tail -12
timeout 300 python bench.py --no-extras 2>&1 | tail -1
} > gpurun_out/sdf3.log 2>&1
