#!/bin/bash
# copies the outputs of tools/gpu_job_r6.sh (gpurun_out/<dir>) into profiles/ under their round-6 names
S=gpurun_out/${1:-r6}; P=profiles
for f in bench_default.json bench_kernel_stats.csv bench_under_rocprof.json pmc_hbm_traffic.json pmc_sq_counters.json \
         bench_virtual_ranks_19.json bench_virtual_ranks_19_kernel_stats.csv pmc_hbm_traffic_virtual_ranks_19.json \
         pmc_sq_counters_virtual_ranks_19.json bench_virtual_ranks_19_one_rank_rccl.json bench_virtual_ranks_19_eight_rows.json \
         bench_virtual_ranks_19_dw64.json bench_virtual_ranks_3_kernel_stats.csv bench_virtual_ranks_3_one_rank_rccl.json \
         bench_num_cpu_19_one_gpu.json bench_two_ranks_gloo_num_cpu_5.json trace_gaps_single.txt cycle_timeline.txt \
         bench_one_rank_rccl_auto.json bench_one_rank_rccl_eager.json bench_two_ranks_gloo_one_gpu.json bench_task_experts.json \
         bench_task_experts_virtual_ranks_3.json bench_task_experts_one_rank_rccl.json bench_arm8_1024env.json \
         rows_stamps_v19.txt rows_stamps_v8.txt rows_stamps_v3.txt rows_stamps_v19_eight_rows.txt dw_timeline_v19.txt; do
  [ -s $S/$f ] && cp $S/$f $P/r06_$f || echo "missing $f"
done
for V in 2 3 4 5 6 7 8 12 16; do cp $S/bench_virtual_ranks_$V.json $P/r06_bench_virtual_ranks_$V.json; done
cp $S/bench_v19_under_rocprof.json $P/r06_bench_virtual_ranks_19_under_rocprof.json
cp $S/bench_v3_under_rocprof.json $P/r06_bench_virtual_ranks_3_under_rocprof.json
cp $S/learn_curious_progress.csv $P/r06_learning_curve_arm4.csv
cp $S/soak_progress.csv $P/r06_soak_arm4_300_epochs.csv
cp $S/learn_num_cpu19_16_progress.csv $P/r06_learning_curve_arm4_num_cpu19_16_rollouts.csv
cp $S/learn_num_cpu19_16_resumed_at_59_progress.csv $P/r06_learning_curve_arm4_num_cpu19_16_rollouts_resumed_at_epoch_59.csv
cat $S/head.txt
