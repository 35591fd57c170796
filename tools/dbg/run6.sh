cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/object-intrinsics_amd/build/ab
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
python tools/parity_margin.py f16x3 2>&1 | tail -1
echo "== default"; python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1 | sed 's/.*"full"/full/'
echo "== prof"; OI_LIB=$AB/liboi_prof.so python tools/dbg/phase_prof3.py 2>&1 | tail -7
