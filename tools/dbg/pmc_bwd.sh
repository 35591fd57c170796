#!/bin/bash
# SQ / LDS / memory counters of the MLP backward kernels (one --pmc pass per group; never combined with other traces)
#   tools/dbg/pmc_bwd.sh <variant|tree>
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
v=${1:-tree}
lib=$PWD/object-intrinsics_amd/build/ab/liboi_$v.so
[ "$v" = tree ] && lib=$PWD/object-intrinsics_amd/oi_amd/liboi_hip.so
{
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc$i
  OI_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_pmc$i -- python tools/dbg/time_bwd.py > /dev/null 2>&1
  python tools/prof_summary.py /tmp/p_pmc$i gpurun_out/pmc_bwd_${v}_$i.txt > /dev/null
  grep -A1 -E "^(mlp_bwd_sweep|mlp_wgrad)" gpurun_out/pmc_bwd_${v}_$i.txt | cut -c1-400
done
} > gpurun_out/pmc_bwd_$v.log 2>&1 < /dev/null
cat gpurun_out/pmc_bwd_$v.log
