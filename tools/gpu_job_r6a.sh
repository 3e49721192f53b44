#!/bin/bash
# Round 6, run A: parity of the 16-row form of the row-local update (csrc/mlp_rows16.h) and its A/B against the 8-row form,
# the round-5 library and the 2-workgroups-per-CU build, interleaved on ONE box.      tools/gpu_job_r6a.sh [outdir]
R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/${1:-r6a}; mkdir -p "$O"
cd "$R" || exit 1
python -m curious_amd.build > /dev/null 2>&1
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k sixteen 2>&1 | tail -15 ) > "$O/tests_r16.txt"
( timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "oracle_rank_model or eight_rows" 2>&1 | tail -8 ) > "$O/tests_r5.txt"
cat "$O/tests_r16.txt" "$O/tests_r5.txt"
run() {  # name, env assignments, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > "$O/$name.json" 2> "$O/$name.err"
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d['kernels']
    g = lambda n: k.get(n, {}).get('avg_us', 0)
    print('%-14s %8.4f ms  %6.2f M/s  rows %.2f rows_her %.2f  dw %.2f' % (sys.argv[1], d['ms_per_step'], d['value'] / 1e6,
          g('ddpg_rows_kernel'), g('ddpg_rows_her_kernel'), g('dw_adam_her_kernel')))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for r in 1 2; do
  run v19_new_$r  "A=1" --virtual-ranks 19
  run v19_r8_$r   "CURIOUS_ROWS16=0" --virtual-ranks 19
  run v19_w2_$r   "CURIOUS_LIB=abtest/r16w2.so" --virtual-ranks 19
  run v19_base_$r "CURIOUS_LIB=abtest/base_r5.so" --virtual-ranks 19
done
run v8_new  "A=1" --virtual-ranks 8
run v8_r8   "CURIOUS_ROWS16=0" --virtual-ranks 8
run v8_w2   "CURIOUS_LIB=abtest/r16w2.so" --virtual-ranks 8
run v4_r16  "CURIOUS_ROWS16=1024" --virtual-ranks 4
run v4_r8   "A=1" --virtual-ranks 4
run v3_r16  "CURIOUS_ROWS16=768" --virtual-ranks 3
run v3_r8   "A=1" --virtual-ranks 3
run v12_new "A=1" --virtual-ranks 12
run v12_r8  "CURIOUS_ROWS16=0" --virtual-ranks 12
for r in 1 2; do
  run v1_new_$r  "A=1"
  run v1_base_$r "CURIOUS_LIB=abtest/base_r5.so"
done
