#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
timeout 120 tools/rows_lab > $O/rows_lab.txt 2>&1; cat $O/rows_lab.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('kernels'))"
