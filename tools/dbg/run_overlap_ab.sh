#!/bin/bash
# same-box A/B of where the two graphed discriminator steps run (OI_TRAIN_D_STEPS: serial | overlap | concurrent), C2 and the
# shipped configuration
cd ${GRAFT_REPO_ROOT:-.}
T="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01 --train-steps 40"
for r in 1 2; do
  for v in serial overlap concurrent; do
    OI_TRAIN_D_STEPS=$v $T 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['training']; print('d_steps=$v', 'ms_per_it', round(t['ms_per_it'],3), 'd_step', round(t['d_step']['ms'],3), 'render_fwd_bwd', round(t['render_fwd_bwd']['ms'],3))"
  done
done
for v in serial concurrent; do
  OI_TRAIN_D_STEPS=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --min-seconds 0.01 --train-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['extras']['training_shipped_config']; print('d_steps=$v shipped config', 'ms_per_it', round(t['ms_per_it'],3), 'it_per_s', round(t['it_per_s'],1))"
done
