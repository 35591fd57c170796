#!/bin/bash
# same-box A/B of the headline step between library variants (build_variants.sh): run_step_ab.sh <variant> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/step_ab.log
for rep in 1 2; do
for v in "$@"; do
  OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so python bench.py --no-cpu-baseline --no-bf16 --no-extras --train-steps ${TRAIN:-0} ${ARGS:-} 2>/dev/null | tail -1 > $O/step_ab_$v.json
  python - $v $O/step_ab_$v.json >> $O/step_ab.log <<'P'
import json,sys
d=json.load(open(sys.argv[2]))
t=d.get('training') or {}
print(sys.argv[1], 'rays/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'train ms/it', t.get('ms_per_it'))
P
done; done
cat $O/step_ab.log
