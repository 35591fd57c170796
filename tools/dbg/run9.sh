cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_bench_fwd3.json
python -c "import json; d=json.load(open('gpurun_out/r2_bench_fwd3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['training'] and d['training']['ms_per_it'])"
cd /tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16 --train-steps 0"
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_$tag; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$tag -- $BENCH > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_$tag $R/gpurun_out/r2_pmc_${tag}_f16x3.txt > /dev/null
  grep -A1 "full3" $R/gpurun_out/r2_pmc_${tag}_f16x3.txt | head -4
done
