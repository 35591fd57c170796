#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_d; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_d -- python $R/tools/dbg/host_dfwd.py > /dev/null 2>&1
python $R/tools/dbg/timeline.py /tmp/p_d /tmp/tl.txt d_aug_conv1_kernel > /dev/null; cat /tmp/tl.txt | cut -c1-120
