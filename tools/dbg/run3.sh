cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
AB=$R/object-intrinsics_amd/build/ab
export TMPDIR=/tmp
echo "== prof"; OI_LIB=$AB/liboi_prof.so python tools/dbg/phase_prof3.py 2>&1 | tail -8
echo "== nopf"; OI_LIB=$AB/liboi_nopf.so python tools/bench_c5.py --modes f16x3 --iters 20 2>&1 | tail -1
echo "== fast trig"; python tools/bench_c5.py --modes f16x3:fast --iters 20 2>&1 | tail -1
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES:sq1" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE:sq2" "GRBM_GUI_ACTIVE:grbm"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_$tag; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$tag -- python $R/tools/bench_c5.py --modes f16x3 --iters 3 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_$tag $R/gpurun_out/r2_c5_pmc_${tag}.txt > /dev/null
  grep -A1 "full3\|sdf_mlp_kernel" $R/gpurun_out/r2_c5_pmc_${tag}.txt | head -12
done
