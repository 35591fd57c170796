#!/bin/bash
# same-box A/B of the f16x3 backward: S_V / S_U slots as 24-bit fixed point (xq1) vs fp32 (xq0): training iteration, gradient margins
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/xq.log
for v in ${VARS:-xq1 xq0 xq1 xq0}; do
  OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so python bench.py --no-cpu-baseline --no-bf16 --no-extras --steps 5 --warmup 2 --min-seconds 0.2 --train-steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['training']; print('$v train ms/it %.4f render fwd+bwd %.4f' % (t['ms_per_it'], t['render_fwd_bwd']['ms']))" >> $O/xq.log
done
for v in ${MVARS:-xq0 xq1}; do
  OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so python tools/grad_margin.py $O/xq_margins_$v.json 2>&1 | grep -E "f16x3|passed|failed" > $O/xq_margins_$v.txt
done
cat $O/xq.log; for v in ${MVARS:-xq0 xq1}; do echo "== $v"; cat $O/xq_margins_$v.txt; done
