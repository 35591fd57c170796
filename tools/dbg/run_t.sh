#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > gpurun_out/t.log 2>&1 < /dev/null
