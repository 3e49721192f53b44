#!/bin/bash
mkdir -p gpurun_out/$1; O=gpurun_out/$1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k=d.get('kernels',{})
    print(sys.argv[1], '%.4f ms'%d['ms_per_step'], '%.2f M/s'%(d['value']/1e6), {n[:9]:v['avg_us'] for n,v in k.items() if n.startswith(('dw_','ddpg_'))})
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for v in 2 3 4; do for cfg in "4096 0" "512 11" "512 12" "4096 11" "4096 12"; do set -- $cfg
  CURIOUS_DW_BAL=$1 CURIOUS_DW_SPLIT=$2 timeout 300 python bench.py --virtual-ranks $v --steps 20 --warmup 5 --no-cpu-baseline > $O/b_v${v}_$1_$2.json 2> $O/b_v${v}_$1_$2.err
  line v${v}_bal$1_split$2 $O/b_v${v}_$1_$2.json
done; done
