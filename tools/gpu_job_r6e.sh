cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r6g}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k sixteen 2>&1 | tail -15 ) > "$O/tests_r16.txt"; cat $O/tests_r16.txt
( timeout 900 python -m pytest tests/test_gpu_round5.py -q -k "oracle_rank_model" 2>&1 | tail -30 ) > "$O/tests_r5.txt"; grep -n "assert dev.max\|Error\|passed\|failed\|^E  " $O/tests_r5.txt | head -30
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
for V in 1 2 3 4 5 6 7 8 9 10 12 16 19; do run v$V "A=1" --virtual-ranks $V; done
run v19_w2 "CURIOUS_LIB=abtest/r16w2.so" --virtual-ranks 19
run v8_w2 "CURIOUS_LIB=abtest/r16w2.so" --virtual-ranks 8
