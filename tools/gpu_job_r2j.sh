#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print('single', d['value'], d['ms_per_step'])"
for g in 0 1; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600+g)) CURIOUS_FORCE_DIST=1 CURIOUS_GRAPH_ALLREDUCE=$g timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_dist$g.json
python -c "import json; d=json.load(open('$O/bench_dist$g.json')); print('one-rank RCCL, captured=$g', d['value'], d['ms_per_step'], {k:(v['launches'], v['avg_us']) for k,v in d['kernels'].items() if k in ('adam_her_kernel','adam_kernel','rows_transpose_kernel','ddpg_rows_kernel','dw_all_kernel')})"
done
