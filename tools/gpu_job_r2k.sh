#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
for m in 1 0; do CURIOUS_ASYNC_STORE=$m timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench async', $m, d['value'], d['ms_per_step'])"; done
cd $O && export PYTHONPATH=$R
for a in 1 0; do
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --n_epochs 150 --n_cycles 25 --n_batches 40 --rollout_batch_size 256 --seed 1 --async_store $a --trial_id $a > learn_async$a.log 2>&1 ) 2> time_async$a.txt
cp save/MultiTaskFetchArm4-v5/$a/progress.csv learn_async${a}_progress.csv
done
rm -rf save
grep real time_async1.txt time_async0.txt
python - <<'PY'
import csv
rows = {}
for a in (1, 0):
    rows[a] = list(csv.DictReader(open('learn_async%d_progress.csv' % a)))
    r = rows[a]
    print('async', a, len(r), [(x['epoch'], x['test/success_rate']) for x in r[::15]], r[-1]['test/success_rate'])
same = all(x == y for x, y in zip(rows[0], rows[1]))
print('progress.csv identical with and without async_store:', same)
if not same:
    for i, (x, y) in enumerate(zip(rows[0], rows[1])):
        if x != y:
            print('first difference at row', i, {k: (x[k], y[k]) for k in x if x[k] != y[k]})
            break
PY
