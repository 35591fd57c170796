cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "graphed" 2>&1 | tail -12
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['training'])"
