cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -5
( time python bench.py ) > gpurun_out/r2_bench_try.json 2> gpurun_out/r2_bench_try.err
tail -3 gpurun_out/r2_bench_try.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_try.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','timing')})
print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
print(d['training'] and {k:d['training'][k] for k in ('it_per_s','ms_per_it')})
print(json.dumps(d.get('extras'))[:3000])
print(d.get('cpu_baseline'))
PY
OI_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.2 --train-steps 3 2>&1 | tail -1 | cut -c1-300
