cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sdf_mlp_golden and f16x3" 2>&1 | grep "f16x3\]\|assert\|Error" | head
