#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -x -q -k "conv4x4 or disc" 2>&1 | tail -3
} > gpurun_out/t.log 2>&1 < /dev/null
