#!/bin/bash
# gpurun driver for the MLP backward: parity tests, then a kernel trace of one C2-size backward (tools/dbg/time_bwd.py)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mlp_backward or f6 or f9 or film_params or c2_size or trajectory or learns" 2>&1 | tail -25
rm -rf gpurun_out/prof_bwd
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -o bwd -- python tools/dbg/time_bwd.py 2>&1 | tail -3
python tools/prof_summary.py gpurun_out/prof_bwd gpurun_out/bwd_stats.txt && head -12 gpurun_out/bwd_stats.txt
} > gpurun_out/bwd.log 2>&1 < /dev/null
tail -60 gpurun_out/bwd.log
