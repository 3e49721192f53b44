cd $GRAFT_REPO_ROOT && python -m curious_amd.build > /dev/null 2>&1
python - <<'PY'
import sys, json, subprocess, os
print('bench.py --steps 20 --warmup 5 --no-cpu-baseline: ms per cycle, kernel averages (us) from the bracketed eager pass')
for env in ({'CURIOUS_ONE_LAUNCH': '0'}, {'CURIOUS_ONE_LAUNCH': '1'}, {'CURIOUS_ONE_LAUNCH': '1', 'CURIOUS_LAB_STEP': '1'}, {'CURIOUS_ONE_LAUNCH': '1', 'CURIOUS_LAB_STEP': '4'}, {'CURIOUS_ONE_LAUNCH': '1', 'CURIOUS_LAB_STEP': '5'}):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline'], capture_output=True, text=True, env=dict(os.environ, **env)).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(env, d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'ddpg' in k or 'dw' in k})
PY
CURIOUS_ONE_LAUNCH=1 python tools/step_stamps.py 2>&1 | tail -14
