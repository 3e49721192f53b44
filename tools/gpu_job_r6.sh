#!/bin/bash
# Round-6 evidence run (HEAD): rocprofv3 kernel stats + PMC passes of bench.py (the headline configuration AND the reference's
# 19-rank regime as virtual ranks), the default bench line (with the CPU baselines), virtual ranks 3 / 8 / 19, --num-cpu 19,
# the several-rank path (one-rank RCCL communicator; 2 ranks on this GPU over gloo incl. --num-cpu 3), batched experts (+ 3
# virtual ranks), configs[2], the phase stamps of the row-local launch, the learning curves, a soak and a resumed job.
#       tools/gpu_job_r6.sh [outdir under gpurun_out/]
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/${1:-r6}; mkdir -p "$O"
cd "$R" || exit 1
python -m curious_amd.build > /dev/null 2>&1
git -C "$R" rev-parse HEAD > "$O/head.txt" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_v19 -- $B --virtual-ranks 19 > $O/bench_v19_under_rocprof.json 2> $O/stats_v19.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_v3 -- $B --virtual-ranks 3 > $O/bench_v3_under_rocprof.json 2> $O/stats_v3.log
S="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $S > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $S > /dev/null 2> $O/pmc_write.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_v19 -- $S --virtual-ranks 19 > /dev/null 2> $O/pmc_fetch_v19.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_v19 -- $S --virtual-ranks 19 > /dev/null 2> $O/pmc_write_v19.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq1 -- $S > /dev/null 2> $O/pmc_sq1.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- $S > /dev/null 2> $O/pmc_sq2.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq1_v19 -- $S --virtual-ranks 19 > /dev/null 2> $O/pmc_sq1_v19.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2_v19 -- $S --virtual-ranks 19 > /dev/null 2> $O/pmc_sq2_v19.log
cd $R
python tools/pmc_summary.py $O/pmc_hbm_traffic.json $O/pmc_fetch $O/pmc_write > $O/pmc_hbm.txt 2>&1
python tools/pmc_summary.py $O/pmc_hbm_traffic_virtual_ranks_19.json $O/pmc_fetch_v19 $O/pmc_write_v19 > $O/pmc_hbm_v19.txt 2>&1
python tools/pmc_summary.py $O/pmc_sq_counters.json $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_sq.txt 2>&1
python tools/pmc_summary.py $O/pmc_sq_counters_virtual_ranks_19.json $O/pmc_sq1_v19 $O/pmc_sq2_v19 > $O/pmc_sq_v19.txt 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/stats_v19 -name "*kernel_stats.csv" -exec cp {} $O/bench_virtual_ranks_19_kernel_stats.csv \;
find $O/stats_v3 -name "*kernel_stats.csv" -exec cp {} $O/bench_virtual_ranks_3_kernel_stats.csv \;
python tools/trace_gaps.py $O/stats --tail 0.6 > $O/trace_gaps_single.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
# the committed PMC passes are what bench.py's roofline.traffic reads: in place before the bench lines are taken
cp $O/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic_virtual_ranks_19.json profiles/r06_pmc_hbm_traffic_virtual_ranks_19.json
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
timeout 400 python bench.py --virtual-ranks 19 > $O/bench_virtual_ranks_19.json 2> $O/bench_virtual_ranks_19.err
for V in 2 3 4 5 6 7 8 12 16; do timeout 200 python bench.py --virtual-ranks $V --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_virtual_ranks_$V.json 2> $O/bench_virtual_ranks_$V.err; done
timeout 200 python bench.py --num-cpu 19 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_num_cpu_19_one_gpu.json 2> $O/bench_num_cpu_19_one_gpu.err
CURIOUS_ROWS16=0 timeout 200 python bench.py --virtual-ranks 19 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_virtual_ranks_19_eight_rows.json 2> /dev/null
CURIOUS_DW64=1280 timeout 200 python bench.py --virtual-ranks 19 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_virtual_ranks_19_dw64.json 2> /dev/null
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_one_rank_rccl_auto.err | grep '^{' > $O/bench_one_rank_rccl_auto.json
CURIOUS_FORCE_DIST=1 CURIOUS_GRAPH_ALLREDUCE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_one_rank_rccl_eager.err | grep '^{' > $O/bench_one_rank_rccl_eager.json
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --structure task_experts --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_task_experts_one_rank_rccl.err | grep '^{' > $O/bench_task_experts_one_rank_rccl.json
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --virtual-ranks 19 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_virtual_ranks_19_one_rank_rccl.err | grep '^{' > $O/bench_virtual_ranks_19_one_rank_rccl.json
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --virtual-ranks 3 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_virtual_ranks_3_one_rank_rccl.err | grep '^{' > $O/bench_virtual_ranks_3_one_rank_rccl.json
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
CURIOUS_DIST_BACKEND=gloo CURIOUS_RESIDENT=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 10 --warmup 3 2> $O/bench_two_ranks_gloo_one_gpu.err | grep '^{' > $O/bench_two_ranks_gloo_one_gpu.json
CURIOUS_DIST_BACKEND=gloo CURIOUS_RESIDENT=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --num-cpu 5 --steps 10 --warmup 3 --no-ipc-probe 2> $O/bench_two_ranks_gloo_num_cpu_5.err | grep '^{' > $O/bench_two_ranks_gloo_num_cpu_5.json
timeout 200 python bench.py --structure task_experts --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_task_experts.json 2> $O/bench_task_experts.err
timeout 200 python bench.py --structure task_experts --virtual-ranks 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_task_experts_virtual_ranks_3.json 2> $O/bench_task_experts_virtual_ranks_3.err
timeout 200 python bench.py --env MultiTaskFetchArm8-v5 --rollout-batch-size 1024 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_arm8_1024env.json 2> $O/bench_arm8_1024env.err
head -c 600 $O/bench_kernel_stats.csv; cat $O/pmc_hbm.txt | tail -20; tail -n 2 $O/*.err
timeout 200 python tools/cycle_timeline.py 2>&1 | grep -v amdgpu.ids > $O/cycle_timeline.txt
# (block stamps of the weight-gradient launch: a build with -DDW_STAMPS, python tools/build_variant.py dwst -DDW_STAMPS)
[ -f abtest/dwst.so ] && CURIOUS_LIB=abtest/dwst.so timeout 200 python tools/dw_timeline.py 19 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/dw_timeline_v19.txt
for V in 19 8 3; do timeout 200 python tools/rows_stamps.py $V 2>&1 | grep -v amdgpu.ids > $O/rows_stamps_v$V.txt; done
CURIOUS_ROWS16=0 timeout 200 python tools/rows_stamps.py 19 2>&1 | grep -v amdgpu.ids > $O/rows_stamps_v19_eight_rows.txt
cd $O && export PYTHONPATH=$R
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --n_epochs 150 --n_cycles 25 --n_batches 40 --rollout_batch_size 256 --seed 1 > learn_curious.log 2>&1 ) 2> time_curious.txt
cp save/MultiTaskFetchArm4-v5/0/progress.csv learn_curious_progress.csv
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --n_epochs 300 --n_cycles 25 --n_batches 100 --rollout_batch_size 256 --seed 3 --trial_id 2 > soak.log 2>&1 ) 2> time_soak.txt
cp save/MultiTaskFetchArm4-v5/2/progress.csv soak_progress.csv
rm -rf "$O/save"
# the reference's rank count on one GPU (virtual ranks, 16-row kernels): 19 x 16 rollouts; killed after 60 epochs and resumed
( time timeout 300 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --num_cpu 19 --rollout_batch_size 16 --n_batches 40 --n_epochs 150 --n_cycles 25 --seed 1 > learn_num_cpu19_16.log 2>&1 ) 2> time_num_cpu19_16.txt
cp save/MultiTaskFetchArm4-v5/0/progress.csv learn_num_cpu19_16_progress.csv
( timeout 300 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --num_cpu 19 --rollout_batch_size 16 --n_batches 40 --n_epochs 60 --n_cycles 25 --seed 1 --trial_id 5 --policy_save_interval 50 > resume_a.log 2>&1 )
( timeout 300 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --num_cpu 19 --rollout_batch_size 16 --n_batches 40 --n_epochs 150 --n_cycles 25 --seed 1 --trial_id 5 --policy_save_interval 50 --resume save/MultiTaskFetchArm4-v5/5 > resume_b.log 2>&1 )
cp save/MultiTaskFetchArm4-v5/5/progress.csv learn_num_cpu19_16_resumed_at_59_progress.csv
ls -la save/MultiTaskFetchArm4-v5/5/training_state > resume_files.txt 2>&1
rm -rf "$O/save"
tail -n 3 time_curious.txt time_soak.txt time_num_cpu19_16.txt
python - <<'PY'
import csv, re
import numpy as np
for f in ('learn_curious_progress.csv', 'soak_progress.csv', 'learn_num_cpu19_16_progress.csv', 'learn_num_cpu19_16_resumed_at_59_progress.csv'):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), [ (r['epoch'], r['test/success_rate']) for r in rows[::max(1,len(rows)//10)] ], rows[-1]['test/success_rate'])
a=list(csv.DictReader(open('learn_num_cpu19_16_progress.csv'))); b=list(csv.DictReader(open('learn_num_cpu19_16_resumed_at_59_progress.csv')))
for r in a + b: r.pop('Time', None)
print('resumed job == uninterrupted job, cell for cell:', a == b, len(a), len(b))
t=np.array([float(m.group(1)) for m in re.finditer(r"over in\s+([0-9.]+)\s+s", open("soak.log").read())])
print(len(t), "epochs of the soak; ms per epoch: first 100 %.1f, last 100 %.1f, median %.1f; epochs over 150 ms: %d" % (1e3*t[:100].mean(), 1e3*t[-100:].mean(), 1e3*np.median(t), int((t > 0.15).sum())))
PY
