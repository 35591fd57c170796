#!/bin/bash
# kernel statistics + PMC traffic of the f16x3 training kernels (sweep, weight-gradient GEMM): gpurun -- 'bash tools/dbg/run_bwd_stats.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TRAIN="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
rm -rf /tmp/q_t; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_t -- $TRAIN --train-steps 20 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/q_t $O/bwd_stats.txt > /dev/null
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/q_$tag; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/q_$tag -- $TRAIN --train-steps 4 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/q_$tag $O/bwd_pmc_$tag.txt > /dev/null
done
grep -E "sweep|wgrad|full3|calls" $O/bwd_stats.txt | cut -c1-140
grep -E "sweep|wgrad" $O/bwd_pmc_fetch.txt | cut -c1-200
grep -E "sweep|wgrad" $O/bwd_pmc_write.txt | cut -c1-200
