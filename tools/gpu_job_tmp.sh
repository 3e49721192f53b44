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
for v in 19 8; do for rep in 1 2; do for lib in base r16nt; do
  L=""; [ $lib = r16nt ] && L=abtest/r16nt.so
  CURIOUS_LIB=$L timeout 300 python bench.py --virtual-ranks $v --steps 20 --warmup 5 --no-cpu-baseline > $O/b${v}_${lib}_$rep.json 2> /dev/null; line v${v}_${lib}_$rep $O/b${v}_${lib}_$rep.json
done; done; done
