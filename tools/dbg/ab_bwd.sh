#!/bin/bash
# same-box A/B of the MLP backward kernels: tools/dbg/ab_bwd.sh <variant> ...   (variant = name under build/ab, or "tree")
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{
for rep in ${REPS:-1 2}; do
for v in "$@"; do
  lib=$PWD/object-intrinsics_amd/build/ab/liboi_$v.so
  [ "$v" = tree ] && lib=$PWD/object-intrinsics_amd/oi_amd/liboi_hip.so
  rm -rf gpurun_out/prof_ab
  OI_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -o ab -- python tools/dbg/time_bwd.py > /dev/null 2>&1
  python tools/prof_summary.py gpurun_out/prof_ab gpurun_out/ab_$v.txt
  echo "== $v (rep $rep)"; grep -E "mlp_bwd_sweep|mlp_wgrad|sdf_mlp_full3" gpurun_out/ab_$v.txt | cut -c1-110
done
done
if [ -f object-intrinsics_amd/build/ab/liboi_bprof.so ]; then
  OI_BWD_SCRATCH_MB=16384 OI_LIB=$PWD/object-intrinsics_amd/build/ab/liboi_bprof.so timeout 300 python tools/dbg/phase_prof_bwd.py 2>&1 | tail -14
fi
} > gpurun_out/ab.log 2>&1 < /dev/null
cat gpurun_out/ab.log
