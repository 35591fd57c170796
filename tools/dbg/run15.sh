cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
cd /tmp
rm -rf /tmp/p_ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.1 --train-steps 10 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks /tmp/ks.txt | grep "bwd_sweep\|wgrad_f16" | cut -c1-110
