cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
T="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/p_ov$v; OI_TRAIN_D_STEPS=$( [ $v = 1 ] && echo overlap || echo serial ) timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ov$v -- $T --train-steps 6 > /dev/null 2>&1
python $R/tools/dbg/overlap_timeline.py /tmp/p_ov$v rows > $R/gpurun_out/t7_ov$v.txt 2>&1
tail -4 $R/gpurun_out/t7_ov$v.txt
done
