#!/bin/bash
# rows-route validation: tests + A/B bench (outputs under gpurun_out/r2d)
O=gpurun_out/r2d; mkdir -p $O
python -m curious_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^|" | grep -v "^---" | tail -40 > $O/pytest.txt
for r in 0 1; do CURIOUS_ROWS=$r timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --phases > $O/bench_rows$r.json 2> $O/bench_rows$r.err; done
tail -n 30 $O/pytest.txt
python - <<'PY'
import json
for r in (0,1):
    try:
        d=json.loads(open('gpurun_out/r2d/bench_rows%d.json'%r).read().strip().splitlines()[-1])
        print(r, d['value'], d['ms_per_step'], d.get('phases'), {k:v['avg_us'] for k,v in d['kernels'].items()})
    except Exception as e: print(r,'ERR',e, open('gpurun_out/r2d/bench_rows%d.err'%r).read()[-800:])
PY
