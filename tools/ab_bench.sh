#!/bin/bash
# A/B of library variants on ONE box: tools/ab_bench.sh OUTDIR "NAME=ENV..." ...  -- every variant is a set of environment
# assignments (e.g. "base=CURIOUS_LIB=abtest/base.so" "xcd=CURIOUS_DW_XCD=1"); the variants run interleaved, ROUNDS times.
O=$1; shift
mkdir -p $O
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs python bench.py --no-cpu-baseline --steps ${STEPS:-40} > $O/${name}_$r.json 2> $O/${name}_$r.err
    python - "$name" "$O/${name}_$r.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d['kernels']
    print('%-10s %.4f ms  rows %.2f  dw %.2f  res %.1f' % (sys.argv[1], d['ms_per_step'],
          k.get('ddpg_rows_kernel', {}).get('avg_us', 0), k.get('dw_adam_her_kernel', {}).get('avg_us', 0),
          k.get('policy_resident_kernel', {}).get('avg_us', 0)))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
  done
done
