#!/bin/bash
# Runs ON the GPU box (gpurun): regenerates every file profiles/ holds for the default (f16x3) mode into gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/r1_* into profiles/.  Counter passes are separate runs (--pmc never combined with other traces).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16"
python $R/bench.py 2>/dev/null | tail -1 > $O/r1_bench_f16x3.json
python $R/bench.py --res 128 --samples 128 --importance 128 --up-steps 4 --steps 10 --warmup 3 --train-steps 0 --no-bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r1_bench_c4_f16x3.json
rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -- $BENCH > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks $O/r1_kernel_stats_f16x3.txt > /dev/null
rm -rf /tmp/p_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- $BENCH --train-steps 0 > /dev/null 2>&1
python $R/tools/dbg/timeline.py /tmp/p_tl $O/r1_timeline_step_f16x3.txt > /dev/null
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE:sq"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_$tag; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$tag -- $BENCH --steps 5 --warmup 2 --train-steps 4 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_$tag $O/r1_pmc_${tag}_f16x3.txt > /dev/null
done
python $R/tools/bench_c5.py > $O/r1_c5_mlp_microbench.jsonl 2>/dev/null
ls -la $O/r1_*
