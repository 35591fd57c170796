#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
rm -rf /tmp/pd; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pd -- python $R/tools/dbg/prof_disc64.py > /dev/null 2>&1 < /dev/null
python $R/tools/prof_summary.py /tmp/pd /tmp/pd.txt < /dev/null > /dev/null 2>&1; head -6 /tmp/pd.txt | cut -c1-150; python $R/tools/dbg/prof_disc64_calls.py /tmp/pd
cd $R; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_backward.py -m gpu -x -q -k "conv4x4 or disc or upfirdn or augment or ada or plugin" 2>&1 | tail -3
} > $R/gpurun_out/t.log 2>&1 < /dev/null
