#!/bin/bash
# samples the shader clock / power (rocm-smi) while the forward and backward MLP kernels run back to back
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
rocm-smi --showclocks --showpower 2>&1 < /dev/null | grep -iE "sclk|power|mclk" | head -6
echo "== under load (backward loop)"
OI_DBG_LOOP=400 timeout 120 python $R/tools/dbg/time_bwd_k.py > /tmp/loop.log 2>&1 < /dev/null &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 < /dev/null | grep -iE "sclk|Average Graphics Package Power|Current Socket" | head -3; sleep 1; done
wait $PID
tail -1 /tmp/loop.log
} > $R/gpurun_out/clock.log 2>&1
