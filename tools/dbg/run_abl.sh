#!/bin/bash
# usage: run_abl.sh <variant> ...   C5 forward timings + the kernel parity tests per library variant
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for v in "$@"; do echo == $v; OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so timeout 300 python $R/tools/bench_c5.py --modes f16x3 2>&1 < /dev/null | grep f16x3 | sed 's/.*"full"/full/' ; done
true
true
} > $R/gpurun_out/abl.log 2>&1
