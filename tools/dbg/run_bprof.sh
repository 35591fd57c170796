#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/dbg/time_bwd.py 2>&1 | tail -1
OI_LIB=$PWD/object-intrinsics_amd/build/ab/liboi_bprof.so timeout 300 python tools/dbg/phase_prof_bwd.py 2>&1 | tail -14
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o pb -- python $GRAFT_REPO_ROOT/tools/dbg/time_bwd.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pb 2>/dev/null | head -8 || find /tmp/pb -name "*kernel_stats*" | head -1 | xargs head -6
} > gpurun_out/bprof.log 2>&1
