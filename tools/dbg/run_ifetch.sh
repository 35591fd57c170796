#!/bin/bash
# instruction-fetch counters of the MLP kernels (is the straight-line 96 KB kernel bound by the 64 KB instruction cache?)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
{
timeout 120 rocprofv3 -L 2>&1 < /dev/null | grep -i -E "ifetch|icache|inst_cache|SQC_|SQ_WAIT|SQ_INSTS_VALU |SQ_INSTS_MFMA|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST" | cut -c1-160 | sort -u | head -80
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40); rm -rf /tmp/q_$tag
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/q_$tag -- python $R/tools/dbg/time_bwd_k.py > /tmp/q_$tag.log 2>&1 < /dev/null
  tail -2 /tmp/q_$tag.log | cut -c1-200
  python $R/tools/prof_summary.py /tmp/q_$tag /tmp/q_$tag.txt < /dev/null > /dev/null 2>&1
  grep -A1 -E "^(mlp_bwd_sweep|mlp_wgrad_f16|sdf_mlp_full3)" /tmp/q_$tag.txt < /dev/null | cut -c1-400
done
} > $R/gpurun_out/ifetch.log 2>&1
