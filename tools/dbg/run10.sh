cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1 | sed 's/.*"full"/full/'
cd /tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16 --train-steps 0"
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_$tag; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$tag -- python $R/tools/bench_c5.py --modes f16x3 --iters 3 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_$tag $R/gpurun_out/r2_c5_pmc_${tag}.txt > /dev/null
  grep -A1 "^sdf_mlp_full3" $R/gpurun_out/r2_c5_pmc_${tag}.txt | head -3
done
