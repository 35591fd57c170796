#!/bin/bash
# kernel-level same-box A/B of the backward between library variants: run_bwd_kab.sh <variant> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TRAIN="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
: > $O/bwd_kab.log
for rep in 1 2; do
for v in "$@"; do
  rm -rf /tmp/q_$v; OI_LIB=$R/object-intrinsics_amd/build/ab/liboi_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$v -- $TRAIN --train-steps 20 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/q_$v $O/bwd_kab_$v.txt > /dev/null
  echo "== $v" >> $O/bwd_kab.log; grep -E "sweep|wgrad_f16" $O/bwd_kab_$v.txt | cut -c1-100 >> $O/bwd_kab.log
done; done
cat $O/bwd_kab.log
