#!/bin/bash
mkdir -p gpurun_out/$1; O=$PWD/gpurun_out/$1
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_agent.py -q -x -k "resum or big_policy or evaluat or checkpoint" 2>&1 | tail -3
R=$PWD; cd $O; export PYTHONPATH=$R
( time timeout 600 python -m curious_amd.experiment.train --env MultiTaskFetchArm4-v5 --n_epochs 160 --n_cycles 25 --n_batches 100 --rollout_batch_size 256 --seed 3 --trial_id 2 > soak.log 2>&1 ) 2> time_soak.txt
python - <<'PY'
import re, numpy as np
t=np.array([float(m.group(1)) for m in re.finditer(r"over in\s+([0-9.]+)\s+s", open("soak.log").read())])
print(len(t), "epochs; ms per epoch: mean %.1f, median %.1f" % (1e3*t.mean(), 1e3*np.median(t)))
print(' '.join('%.0f' % (1e3*x) for x in t))
PY
tail -3 time_soak.txt
ls -la save/MultiTaskFetchArm4-v5/2/training_state/ | head
rm -rf save
