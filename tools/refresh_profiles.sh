#!/bin/bash
# Runs ON the GPU box (gpurun): regenerates every file profiles/ holds for the default (f16x3) mode into gpurun_out/.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/r6_* into profiles/ and run tools/profiles_index.py r6 --write.  Counter passes are separate runs (--pmc never combined with other traces).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${OI_PROFILE_TAG:-r6}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# every profiled run under its own timeout: one hung step (seen once: the one-rank process-group run never returned) must not
# take the whole refresh -- and the GPU budget -- with it
RP="timeout 420 rocprofv3"
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.2"
# (forward steps only: the with-gradient renders of the training leg run the same kernel with the feature stores on, and would
#  pull its average away from what bench.py's HIP events measure; the training kernels are in ${T}_kernel_stats_train.txt)
rm -rf /tmp/p_ks; $RP --kernel-trace --stats --output-format csv -d /tmp/p_ks -- $BENCH --train-steps 0 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_ks $O/${T}_kernel_stats_f16x3.txt > /dev/null
rm -rf /tmp/p_tl; $RP --kernel-trace --output-format csv -d /tmp/p_tl -- $BENCH --train-steps 0 > /dev/null 2>&1
python $R/tools/dbg/timeline.py /tmp/p_tl $O/${T}_timeline_step_f16x3.txt prep_render_kernel > /dev/null
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE:sq"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_$tag; $RP --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$tag -- $BENCH --steps 5 --warmup 2 --train-steps 4 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_$tag $O/${T}_pmc_${tag}_f16x3.txt > /dev/null
done
# training iteration: kernel stats, and launches per iteration from the difference of two runs (10 vs 30 iterations)
TRAIN="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
rm -rf /tmp/p_t10 /tmp/p_t30
$RP --kernel-trace --output-format csv -d /tmp/p_t10 -- $TRAIN --train-steps 10 > /dev/null 2>&1
$RP --kernel-trace --output-format csv -d /tmp/p_t30 -- $TRAIN --train-steps 30 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_t30 $O/${T}_kernel_stats_train.txt > /dev/null
python $R/tools/train_launches.py /tmp/p_t10 10 /tmp/p_t30 30 > $O/${T}_timeline_train.txt
# RCCL on a 1-GPU box: the same job with a process group of one rank (FlatGradDDP then launches its all-reduces)
rm -rf /tmp/p_dist; OI_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 $RP --kernel-trace --stats --output-format csv -d /tmp/p_dist -- $TRAIN --train-steps 10 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_dist $O/${T}_kernel_stats_train_rccl_1rank.txt > /dev/null
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_f16x3.txt $O/${T}_pmc_write_f16x3.txt sdf_mlp_full3_kernel "f16x3:1x64x64:64+64" $O/${T}_traffic.json
# the bf16 mode's dominant kernel (sdf_mlp_full3p_kernel<true>: register-resident, per-element images, fast trig) + its kernel stats / timeline
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_b$tag; $RP --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_b$tag -- $BENCH --precision bf16 --steps 5 --warmup 2 --train-steps 0 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_b$tag $O/${T}_pmc_${tag}_bf16.txt > /dev/null
done
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_bf16.txt $O/${T}_pmc_write_bf16.txt "sdf_mlp_full3p_kernel" "bf16:1x64x64:64+64" $O/${T}_traffic.json
rm -rf /tmp/p_bks; $RP --kernel-trace --stats --output-format csv -d /tmp/p_bks -- $BENCH --precision bf16 --train-steps 0 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_bks $O/${T}_kernel_stats_bf16.txt > /dev/null
python $R/tools/dbg/timeline.py /tmp/p_bks $O/${T}_timeline_step_bf16.txt prep_render_kernel > /dev/null
rm -rf /tmp/p_bsq; $RP --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_bsq -- $BENCH --precision bf16 --steps 5 --warmup 2 --train-steps 0 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_bsq $O/${T}_pmc_sq_bf16.txt > /dev/null
# bf16 mode END TO END (BASELINE configs[1]): the training iteration with the generator in the bf16 operand mode -- kernel stats
# and the PMC traffic of its backward (mlp_bwd_sweep_kernel<2, ...> + mlp_wgrad_kernel)
TRAINB="python $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-extras --min-seconds 0.01"
rm -rf /tmp/p_tb; $RP --kernel-trace --output-format csv -d /tmp/p_tb -- $TRAINB --train-steps 20 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_tb $O/${T}_kernel_stats_train_bf16.txt > /dev/null
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_tb$tag; $RP --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_tb$tag -- $TRAINB --train-steps 4 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_tb$tag $O/${T}_pmc_${tag}_train_bf16.txt > /dev/null
done
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_train_bf16.txt $O/${T}_pmc_write_train_bf16.txt "mlp_bwd_sweep_kernel<2" "bf16-train-sweep:1x64x64:64+64" $O/${T}_traffic.json
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_train_bf16.txt $O/${T}_pmc_write_train_bf16.txt "mlp_wgrad_f16_kernel<true, true" "bf16-train-wgrad:1x64x64:64+64" $O/${T}_traffic.json
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_f16x3.txt $O/${T}_pmc_write_f16x3.txt "mlp_bwd_sweep_kernel<4" "f16x3-train-sweep:1x64x64:64+64" $O/${T}_traffic.json
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_f16x3.txt $O/${T}_pmc_write_f16x3.txt "mlp_wgrad_f16_kernel" "f16x3-train-wgrad:1x64x64:64+64" $O/${T}_traffic.json
# the batch-64 discriminator forward (csrc/disc_large.hip): one step's timeline + kernel stats
rm -rf /tmp/p_d64; $RP --kernel-trace --stats --output-format csv -d /tmp/p_d64 -- python $R/tools/dbg/prof_disc64.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/p_d64 $O/${T}_kernel_stats_disc_b64.txt > /dev/null
python $R/tools/dbg/timeline.py /tmp/p_d64 $O/${T}_timeline_disc_b64.txt ada_sep_kernel > /dev/null
# the stand-alone albedo head (oi_color_head_fwd / _bwd) at the C2 point count
timeout 300 python $R/tools/dbg/time_color_head.py > $O/${T}_color_head.txt 2>/dev/null
# the same dominant kernel at the C4 per-GPU size (128^2 rays, 128 + 128 samples, 4 up-sampling steps: eight times the points)
C4="--res 128 --samples 128 --importance 128 --up-steps 4"
for c in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  ctr=${c%%:*}; tag=${c##*:}
  rm -rf /tmp/p_c$tag; $RP --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_c$tag -- $BENCH $C4 --steps 3 --warmup 1 --train-steps 0 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/p_c$tag $O/${T}_pmc_${tag}_c4.txt > /dev/null
done
python $R/tools/traffic_json.py $O/${T}_pmc_fetch_c4.txt $O/${T}_pmc_write_c4.txt sdf_mlp_full3_kernel "f16x3:1x128x128:128+128" $O/${T}_traffic.json
timeout 300 python $R/tools/bench_c5.py > $O/${T}_c5_mlp_microbench.jsonl 2>/dev/null
timeout 600 python $R/tools/grad_margin.py $O/${T}_margins.json > $O/${T}_gradient_margins.txt 2>/dev/null   # (incl. the bf16-mode map margins)
# the un-profiled bench lines last: they read the traffic file written above (same sources, same digest)
cp $O/${T}_traffic.json $R/profiles/${T}_traffic.json
timeout 900 python $R/bench.py 2>/dev/null | tail -1 > $O/${T}_bench_f16x3.json
timeout 600 python $R/bench.py --res 128 --samples 128 --importance 128 --up-steps 4 --steps 10 --warmup 3 --train-steps 0 --no-bf16 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${T}_bench_c4_f16x3.json
ls -la $O/${T}_*
