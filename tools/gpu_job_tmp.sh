cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r6u; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_agent.py tests/test_gpu_round3.py -x -q 2>&1 | tail -4 ) 
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
for r in 1 2; do
run v1_new_$r "A=1"
run v1_old_$r "CURIOUS_LIB=abtest/base_r5.so"
done
run v19 "A=1" --virtual-ranks 19
run v8 "A=1" --virtual-ranks 8
run v3 "A=1" --virtual-ranks 3
run arm8 "A=1" --env MultiTaskFetchArm8-v5 --rollout-batch-size 1024
