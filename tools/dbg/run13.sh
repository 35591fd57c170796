cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['training'])"
