#!/bin/bash
# usage: run_st.sh <variant> ...   kernel times of one C2-size backward per library variant (build/ab/liboi_<variant>.so)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
for v in "$@"; do echo == $v; rm -rf /tmp/p_$v
  OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_${v%_fast}.so OI_DBG_FAST=$( [ "${v%_fast}" != "$v" ] && echo 1 ) timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$v -- python $R/tools/dbg/time_bwd_k.py 2>&1 < /dev/null | grep backward
  python $R/tools/prof_summary.py /tmp/p_$v /tmp/p_$v.txt < /dev/null > /dev/null 2>&1; grep -E "sweep|wgrad" /tmp/p_$v.txt < /dev/null | cut -c1-100
done
} > $R/gpurun_out/st.log 2>&1
