#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
cd $R; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
} > $R/gpurun_out/final_tests.log 2>&1 < /dev/null
bash $R/tools/refresh_profiles.sh > $R/gpurun_out/refresh.log 2>&1 < /dev/null
