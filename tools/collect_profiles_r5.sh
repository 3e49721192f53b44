#!/bin/bash
# copies the outputs of tools/gpu_job_r5.sh (gpurun_out/<dir>) into profiles/ under their round-5 names
S=gpurun_out/${1:-r5a}; P=profiles
for f in bench_default.json bench_kernel_stats.csv bench_under_rocprof.json pmc_hbm_traffic.json pmc_sq_counters.json \
         bench_virtual_ranks_19.json bench_virtual_ranks_3.json bench_virtual_ranks_19_kernel_stats.csv \
         pmc_hbm_traffic_virtual_ranks_19.json pmc_sq_counters_virtual_ranks_19.json bench_virtual_ranks_19_one_rank_rccl.json \
         trace_gaps_single.txt cycle_timeline.txt bench_one_rank_rccl_auto.json bench_one_rank_rccl_eager.json \
         bench_two_ranks_gloo_one_gpu.json bench_task_experts.json bench_task_experts_one_rank_rccl.json bench_arm8_1024env.json; do
  [ -s $S/$f ] && cp $S/$f $P/r05_$f || echo "missing $f"
done
cp $S/bench_v19_under_rocprof.json $P/r05_bench_virtual_ranks_19_under_rocprof.json
cp $S/learn_curious_progress.csv $P/r05_learning_curve_arm4.csv
cp $S/soak_progress.csv $P/r05_soak_arm4_300_epochs.csv
cp $S/floor2_lab.txt $P/r05_floor2_lab.txt
cp $S/learn_num_cpu19_16_progress.csv $P/r05_learning_curve_arm4_num_cpu19_16_rollouts.csv
[ -s $S/learn_num_cpu19_ref_progress.csv ] && cp $S/learn_num_cpu19_ref_progress.csv $P/r05_learning_curve_arm4_num_cpu19_reference_regime_800_epochs.csv
cat $S/head.txt
