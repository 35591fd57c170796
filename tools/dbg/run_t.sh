#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "wide_dynamic" 2>&1 | tail -15
} > gpurun_out/t.log 2>&1 < /dev/null
