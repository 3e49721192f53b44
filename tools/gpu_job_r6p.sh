cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r6p}; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -k "bench or resumes_bit" 2>&1 | tail -25 ) > $O/t.txt; grep -n "passed\|failed\|^E " $O/t.txt | head -20
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --num-cpu 19 > $O/b_numcpu19.json 2> $O/b_numcpu19.err; python -c "
import json; d=json.load(open('$O/b_numcpu19.json')); print(d['value'], d['ms_per_step'], d['config']['ranks'], d['config']['ranks_per_process'])"
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --structure task_experts --virtual-ranks 3 > $O/b_experts_v3.json 2> $O/b_experts_v3.err; tail -3 $O/b_experts_v3.err; python -c "
import json; d=json.load(open('$O/b_experts_v3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['workload'][-200:])"
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --structure task_experts > $O/b_experts.json 2> $O/b_experts.err; python -c "
import json; d=json.load(open('$O/b_experts.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
