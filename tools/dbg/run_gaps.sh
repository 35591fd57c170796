#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
rm -rf /tmp/p_tg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tg -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01 --train-steps 24 > /dev/null 2>&1 < /dev/null
python $R/tools/dbg/train_gaps.py /tmp/p_tg 8 8 < /dev/null
} > $R/gpurun_out/gaps.log 2>&1
