#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.2"
rm -rf /tmp/p_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- $BENCH --train-steps 0 > /dev/null 2>&1
python $R/tools/dbg/timeline.py /tmp/p_tl $R/gpurun_out/tl_step.txt prep_render_kernel > /dev/null
cat $R/gpurun_out/tl_step.txt | cut -c1-150
