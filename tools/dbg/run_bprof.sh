#!/bin/bash
mkdir -p gpurun_out
{
OI_BWD_SCRATCH_MB=16384 OI_LIB=$PWD/object-intrinsics_amd/build/ab/liboi_bprof.so timeout 300 python tools/dbg/phase_prof_bwd.py 2>&1 | tail -14
} > gpurun_out/bprof.log 2>&1 < /dev/null
