#!/bin/bash
# launches per training iteration (difference of a 10- and a 30-iteration trace): gpurun_out/train_launches.txt
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=$PWD
TRAIN="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
rm -rf /tmp/p_t10 /tmp/p_t30
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_t10 -- $TRAIN --train-steps 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_t30 -- $TRAIN --train-steps 30 > /dev/null 2>&1
python $R/tools/train_launches.py /tmp/p_t10 10 /tmp/p_t30 30 > gpurun_out/train_launches.txt
head -${1:-70} gpurun_out/train_launches.txt | cut -c1-140
