cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r6d}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k sixteen 2>&1 | tail -15 ) > "$O/tests_r16.txt"; cat $O/tests_r16.txt
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
run v19 "A=1" --virtual-ranks 19
run v19_r8 "CURIOUS_ROWS16=0" --virtual-ranks 19
run v12 "A=1" --virtual-ranks 12
run v10 "A=1" --virtual-ranks 10
run v10_x0 "CURIOUS_ROWS_XCD=0" --virtual-ranks 10
run v8 "A=1" --virtual-ranks 8
run v8_r8 "CURIOUS_ROWS16=0" --virtual-ranks 8
run v6_r16 "CURIOUS_ROWS16=1536" --virtual-ranks 6
run v6_r8 "A=1" --virtual-ranks 6
python tools/rows_stamps.py 19 > $O/stamps_v19.txt 2>&1; tail -n 7 $O/stamps_v19.txt
python tools/rows_stamps.py 8 > $O/stamps_v8.txt 2>&1; tail -n 7 $O/stamps_v8.txt
