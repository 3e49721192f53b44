cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r6h}; mkdir -p $O
run() { local name=$1 envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > "$O/$name.json" 2> "$O/$name.err"
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); k = d['kernels']; g = lambda n: k.get(n, {}).get('avg_us', 0)
    print('%-14s %8.4f ms  %6.2f M/s  rows %.2f  dw %.2f' % (sys.argv[1], d['ms_per_step'], d['value'] / 1e6, g('ddpg_rows_kernel'), g('dw_adam_her_kernel')))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for sp in 0 14 24 34 44 22 12 18 28; do run v19_s$sp "CURIOUS_DW_SPLIT=$sp" --virtual-ranks 19; done
for sp in 0 22 24; do run v19_x0_s$sp "CURIOUS_DW_SPLIT=$sp CURIOUS_DW_XCD=0" --virtual-ranks 19; done
for sp in 0 14 24 22; do run v8_s$sp "CURIOUS_DW_SPLIT=$sp" --virtual-ranks 8; done
( timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/full.txt; tail -5 $O/full.txt
