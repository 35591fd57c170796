#!/bin/bash
# same-box A/B of library variants on the captured discriminator step: tools/dbg/run_dstep_ab.sh <variant> ... ("tree" = the built library)
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/object-intrinsics_amd/build/ab/liboi_$v.so
  [ "$v" = tree ] && lib=$PWD/object-intrinsics_amd/oi_amd/liboi_hip.so
  OI_LIB=$lib python bench.py --no-cpu-baseline --no-bf16 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['training']
print('$v', round(t['ms_per_it'],3), round(t['d_step']['ms'],4), round(t['d_step']['eager_ms'],3), round(t['render_fwd_bwd']['ms'],3))"
done; done
