#!/bin/bash
# same-box A/B of the side-stream discriminator step (OI_TRAIN_OVERLAP, OI_TRAIN_OVERLAP_PRIO)
cd ${GRAFT_REPO_ROOT:-.}
T="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01 --train-steps 40"
for r in 1 2; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    OI_TRAIN_OVERLAP=$1 OI_TRAIN_OVERLAP_PRIO=$2 $T 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['training']; print('overlap=$1 prio=$2', 'ms_per_it', round(t['ms_per_it'],3), 'd_step', round(t['d_step']['ms'],3), 'render_fwd_bwd', round(t['render_fwd_bwd']['ms'],3))"
  done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_ov; OI_TRAIN_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ov -- $T --train-steps 6 > /dev/null 2>&1
f=$(ls /tmp/p_ov/*/*kernel_trace.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# overlap statistics: for the last 400 kernels, print start, dur, name, stream/queue
t0=int(rows[-420]['Start_Timestamp'])
for r in rows[-420:-200]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:60]}")
PY
