#!/bin/bash
# kernel times of the batch-1 discriminator forward (rocprofv3 stats) + un-profiled wall time per image
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_d; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_d -- python $R/tools/dbg/host_dfwd.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_d /tmp/ds.txt > /dev/null; grep -E "d_aug|d_conv|ada_pad" /tmp/ds.txt | cut -c1-120
cd $R; python tools/dbg/host_dfwd.py 2>&1 | grep " us" | head -1
