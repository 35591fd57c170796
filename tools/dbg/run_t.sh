#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/dbg/time_bwd_k.py 2>&1 | tail -1
} > gpurun_out/t.log 2>&1 < /dev/null
