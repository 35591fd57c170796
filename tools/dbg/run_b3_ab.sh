#!/bin/bash
# A/B of bf16 forward-kernel variants on ONE box: tools/dbg/build_variants.sh mlp_fwd3b.hip <name> "<flags>" ... first, then
#   gpurun -- 'bash tools/dbg/run_b3_ab.sh base name1 name2 ...'   (base = the in-tree library)
cd $(dirname $0)/../..
for v in "$@"; do
  lib=object-intrinsics_amd/build/ab/liboi_$v.so
  [ "$v" = "base" ] && lib=object-intrinsics_amd/oi_amd/liboi_hip.so
  echo "== $v"
  OI_LIB=$PWD/$lib python tools/bench_c5.py --modes bf16 --iters 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   full %.4f ms  sdf-only %.4f ms' % (d['full']['ms'], d['sdf_only']['ms']))"
done
