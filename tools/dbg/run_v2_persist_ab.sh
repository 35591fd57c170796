#!/bin/bash
# same-box A/B of the sdf-only (v2) passes: persistent workgroups (default) vs OI_V2_PERSIST=0, + a bit-for-bit comparison
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3) > $O/v2p_tests.log
: > $O/v2p.log
for m in f16x3 bf16 f32; do
OI_V2_PERSIST=0 python tests/helpers/fwd_dump.py $O/fwd_p0.pt $m > /dev/null 2>&1
OI_V2_PERSIST=1 python tests/helpers/fwd_dump.py $O/fwd_p1.pt $m > /dev/null 2>&1
echo "-- $m" >> $O/v2p.log; python tests/helpers/fwd_dump.py --cmp $O/fwd_p0.pt $O/fwd_p1.pt | grep sdf0 >> $O/v2p.log 2>&1
done
rm -f $O/fwd_p0.pt $O/fwd_p1.pt
for v in 1 0 1 0; do
  echo "persist=$v" >> $O/v2p.log
  OI_V2_PERSIST=$v python tools/bench_c5.py --modes f16x3,bf16 2>&1 | grep '"mode"' | cut -c1-200 >> $O/v2p.log
  OI_V2_PERSIST=$v python bench.py --no-cpu-baseline --no-extras --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['bf16_mode']; print('  step: f16x3 rays/s %.0f ms/step %.4f | bf16 rays/s %.0f ms/step %.4f' % (d['value'], d['ms_per_step'], b['value'], b['ms_per_step']))" >> $O/v2p.log
done
for v in 1 0; do
  OI_V2_PERSIST=$v python bench.py --res 128 --samples 128 --importance 128 --up-steps 4 --steps 10 --warmup 3 --train-steps 0 --no-bf16 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  C4 persist=$v: rays/s %.0f ms/step %.4f' % (d['value'], d['ms_per_step']))" >> $O/v2p.log
done
cat $O/v2p_tests.log $O/v2p.log
