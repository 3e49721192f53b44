#!/bin/bash
# copies the outputs of tools/gpu_job_r4.sh (gpurun_out/<dir>) into profiles/ under their round-4 names
S=gpurun_out/${1:-r4y}; P=profiles
for f in bench_default.json bench_kernel_stats.csv bench_under_rocprof.json pmc_hbm_traffic.json pmc_sq_counters.json dw_stamps.txt \
         trace_gaps_single.txt cycle_timeline.txt bench_one_rank_rccl_auto.json bench_one_rank_rccl_eager.json \
         bench_two_ranks_gloo_one_gpu.json bench_task_experts.json bench_task_experts_one_rank_rccl.json bench_arm8_1024env.json \
         bench_resident_off.json bench_dw_xcd_off.json; do
  [ -s $S/$f ] && cp $S/$f $P/r04_$f || echo "missing $f"
done
cp $S/learn_curious_progress.csv $P/r04_learning_curve_arm4.csv
cp $S/learn_experts_progress.csv $P/r04_learning_curve_arm4_task_experts_batched_normalize_obs.csv
cp $S/soak_progress.csv $P/r04_soak_arm4_300_epochs.csv
cat $S/head.txt
