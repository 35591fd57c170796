cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py -x -q -m gpu -k "f10 or f11 or c2_size" 2>&1 | tail -15
