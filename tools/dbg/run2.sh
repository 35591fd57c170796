cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/object-intrinsics_amd/build/ab
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8
python tools/parity_margin.py f16x3 2>&1 | tail -1
echo "== default"; python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1
for v in nogroups v12 v7; do echo "== $v"; OI_LIB=$AB/liboi_$v.so python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1; done
echo "== v2"; OI_FWD_V2=1 python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16 --train-steps 0 2>&1 | tail -1
