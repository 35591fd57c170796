cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['training'])"
