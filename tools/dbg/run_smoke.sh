#!/bin/bash
mkdir -p gpurun_out
{
time (timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3)
time (timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json)
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline'].get('hbm'), d['training']['it_per_s'], d['training']['d_step']['ms'])
PY
} > gpurun_out/smoke.log 2>&1 < /dev/null
