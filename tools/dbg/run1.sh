set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15
python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -2
OI_FWD_V2=1 python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -2
python tools/parity_margin.py f16x3 f32 2>&1 | tail -3
tools/dbg/bin/atomic_probe 2>&1 | tail -15
tools/dbg/bin/tr_probe 2>&1 | head -40
