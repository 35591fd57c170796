#!/bin/bash
# usage: run_fwd_ab.sh <variant> ...   per library variant (build_variants.sh): C5 timings (2^21 points, sdf-only and
# full pass, accurate and fast trig) and a seeded forward whose outputs are compared bit for bit with the FIRST variant's
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
first=""
{
for v in "$@"; do
  echo == $v
  export OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so
  timeout 300 python $R/tools/bench_c5.py --modes ${MODES:-f16x3,f16x3:fast} 2>&1 < /dev/null | grep '"mode"'
  timeout 300 python $R/tests/helpers/fwd_dump.py $R/gpurun_out/fwd_$v.pt 2>&1 | tail -1
  if [ -z "$first" ]; then first=$v; else python $R/tests/helpers/fwd_dump.py --cmp $R/gpurun_out/fwd_$first.pt $R/gpurun_out/fwd_$v.pt; fi
done
rm -f $R/gpurun_out/fwd_*.pt
true
} > $R/gpurun_out/fwd_ab.log 2>&1
