#!/bin/bash
# same-box A/B of the film context (OI_FILM_CTX=0: per-call builders / in-kernel staging), both operand modes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ctx_ab.log
for rep in 1 2; do
for v in 0 1; do
  OI_FILM_CTX=$v python bench.py --no-cpu-baseline --no-extras --train-steps ${TRAIN:-10} ${ARGS:-} 2>/dev/null | tail -1 > $O/ctx_ab_$v.json
  python - $v $O/ctx_ab_$v.json >> $O/ctx_ab.log <<'P'
import json,sys
d=json.load(open(sys.argv[2])); b=d.get('bf16_mode') or {}; t=d.get('training') or {}
print('ctx=%s' % sys.argv[1], 'f16x3: rays/s %.0f ms/step %.4f kernel_ms %.4f train %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], t.get('ms_per_it', 0)),
      '| bf16: rays/s %.0f ms/step %.4f kernel_ms %.4f train %.3f' % (b.get('value',0), b.get('ms_per_step',0), b.get('kernel_ms',0), (b.get('training') or {}).get('ms_per_it',0)))
P
done; done
cat $O/ctx_ab.log
