cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r6n}; mkdir -p $O
for v in dwst_ns dwst; do
echo "== $v"; CURIOUS_LIB=abtest/$v.so python tools/dw_timeline.py 19 > $O/tl_$v.txt 2>&1; tail -5 $O/tl_$v.txt
done
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "64x64" 2>&1 | tail -5 ) > "$O/tests_r6.txt"; cat $O/tests_r6.txt
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
for V in 19 12 8 5; do
  run v${V}_dw64 "A=1" --virtual-ranks $V
  run v${V}_dw16 "CURIOUS_DW64=0" --virtual-ranks $V
done
