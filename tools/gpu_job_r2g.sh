#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --phases 2>/dev/null | grep '^{' > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['phases'])"
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
