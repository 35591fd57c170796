#!/bin/bash
# sweep / wgrad kernel time vs number of points: 64, 128, 256 workgroups = one tile on a quarter / half / all of the CUs
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
for n in 8192 16384 32768 65536 131072 524288; do echo == n=$n; rm -rf /tmp/p_$n
  OI_DBG_N=$n timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$n -- python $R/tools/dbg/time_bwd_k.py 2>&1 < /dev/null | grep backward
  python $R/tools/prof_summary.py /tmp/p_$n /tmp/p_$n.txt < /dev/null > /dev/null 2>&1; grep -E "sweep|wgrad|full3" /tmp/p_$n.txt < /dev/null | cut -c1-100
done
} > $R/gpurun_out/scale.log 2>&1
