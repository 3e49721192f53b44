#!/bin/bash
# Round-3 profiles: rocprofv3 kernel stats + PMC passes of bench.py, default bench line (with the CPU baselines), the
# several-rank path on a one-rank RCCL communicator, batched experts (one rank / one-rank RCCL), configs[2], labs, curves.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.log
S="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $S > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $S > /dev/null 2> $O/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq1 -- $S > /dev/null 2> $O/pmc_sq1.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- $S > /dev/null 2> $O/pmc_sq2.log
cd $R
python tools/pmc_summary.py $O/pmc_hbm_traffic.json $O/pmc_fetch $O/pmc_write > $O/pmc_hbm.txt 2>&1
python tools/pmc_summary.py $O/pmc_sq_counters.json $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_sq.txt 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
python tools/trace_gaps.py $O/stats --tail 0.6 > $O/trace_gaps_single.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_one_rank_rccl_auto.err | grep '^{' > $O/bench_one_rank_rccl_auto.json
CURIOUS_FORCE_DIST=1 CURIOUS_GRAPH_ALLREDUCE=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_one_rank_rccl_eager.err | grep '^{' > $O/bench_one_rank_rccl_eager.json
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --structure task_experts --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_task_experts_one_rank_rccl.err | grep '^{' > $O/bench_task_experts_one_rank_rccl.json
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
timeout 200 python bench.py --structure task_experts --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_task_experts.json 2> $O/bench_task_experts.err
timeout 200 python bench.py --env MultiTaskFetchArm8-v5 --rollout-batch-size 1024 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_arm8_1024env.json 2> $O/bench_arm8_1024env.err
CURIOUS_RESIDENT=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_resident_off.json 2> $O/bench_resident_off.err
head -c 600 $O/bench_kernel_stats.csv; cat $O/pmc_hbm.txt | tail -20; tail -n 2 $O/*.err
timeout 200 python tools/cycle_timeline.py 2>&1 | grep -v amdgpu.ids > $O/cycle_timeline.txt
( timeout 200 python tools/exploit_cycle_probe.py --cycles 200; timeout 200 python tools/exploit_cycle_probe.py --cycles 200 --no-freeze; timeout 200 python tools/exploit_cycle_probe.py --cycles 200 --env MultiTaskFetchArm8-v5 --rollout-batch-size 1024 ) 2>&1 | grep -v amdgpu.ids > $O/cycle_probe.txt
( timeout 100 tools/rows_lab 256; timeout 100 tools/rows_lab 512 | grep '^B\|per launch'; timeout 100 tools/rows_lab 64 | grep '^B\|per launch\|actor side  (sh' ) > $O/rows_lab.txt 2>&1
cd $O && export PYTHONPATH=$R
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --n_epochs 150 --n_cycles 25 --n_batches 40 --rollout_batch_size 256 --seed 1 > learn_curious.log 2>&1 ) 2> time_curious.txt
cp save/MultiTaskFetchArm4-v5/0/progress.csv learn_curious_progress.csv
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --structure task_experts --task_selection random --task_replay replay_current_task_buffer --experts_update batched --n_epochs 80 --n_cycles 25 --n_batches 40 --rollout_batch_size 256 --seed 1 --trial_id 1 > learn_experts.log 2>&1 ) 2> time_experts.txt
cp save/MultiTaskFetchArm4-v5/1/progress.csv learn_experts_progress.csv
rm -rf save
tail -n 3 time_curious.txt time_experts.txt
python - <<'PY'
import csv
for f in ('learn_curious_progress.csv','learn_experts_progress.csv'):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), [ (r['epoch'], r['test/success_rate']) for r in rows[::max(1,len(rows)//10)] ], rows[-1]['test/success_rate'])
PY
