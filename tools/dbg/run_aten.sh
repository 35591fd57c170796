#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python tools/dbg/prof_iter_aten.py 2>&1 | tail -40 | cut -c1-160
