#!/bin/bash
mkdir -p gpurun_out/att
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
rocprofv3 --help 2>&1 < /dev/null | grep -i -A3 "att" | head -60
rm -rf /tmp/att
timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "full3" -d /tmp/att -- python $R/tools/dbg/att_fwd.py 2>&1 < /dev/null | tail -15
find /tmp/att -type f | head -40
du -sh /tmp/att
} > $R/gpurun_out/att.log 2>&1
