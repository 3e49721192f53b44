#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
python -m curious_amd.build > /dev/null 2>&1
echo skip-tests > $O/pytest.txt
tail -n 5 $O/pytest.txt
for v in auto 0; do
  if [ $v = auto ]; then unset CURIOUS_GRAPH_ALLREDUCE; else export CURIOUS_GRAPH_ALLREDUCE=$v; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29000 + RANDOM % 900)) CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/fd_$v.err | grep '^{' > $O/fd_$v.json; echo "rc=$?" >> $O/fd_$v.err
done
unset CURIOUS_GRAPH_ALLREDUCE
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --phases > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ('fd_auto','fd_0','bench'):
    try:
        d=json.loads(open('gpurun_out/r2f/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('phases'))
    except Exception as e: print(f,'ERR',e)
PY
tail -n 3 $O/fd_auto.err
