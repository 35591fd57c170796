#!/bin/bash
# same-box A/B of the bf16 forward kernel: persistent workgroups (default) vs OI_B3P_PERSIST=0 (one workgroup per tile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py -m gpu -x -q -k "bf16 or golden or ragged" 2>&1 | tail -4) > $O/b3pp_tests.log
: > $O/b3pp.log
for v in 1 0 1 0; do
  echo "persist=$v" >> $O/b3pp.log
  OI_B3P_PERSIST=$v python tools/bench_c5.py --modes bf16 2>&1 | grep '"mode"' >> $O/b3pp.log
  OI_B3P_PERSIST=$v python bench.py --precision bf16 --no-cpu-baseline --no-bf16 --no-extras --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  step: rays/s %.0f ms/step %.4f kernel_ms %.4f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" >> $O/b3pp.log
done
cat $O/b3pp_tests.log $O/b3pp.log
