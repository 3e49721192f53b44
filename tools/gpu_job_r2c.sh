#!/bin/bash
# round-2 GPU job: tests + bench variants (outputs under gpurun_out/r2c)
O=gpurun_out/r2c; mkdir -p $O
python -m curious_amd.build > /dev/null 2>&1
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "^|" | grep -v "^---" | tail -15 > $O/pytest.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
CURIOUS_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "rc=$?" >> $O/bench_forcedist.err
timeout 200 python bench.py --structure task_experts --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_experts.json 2> $O/bench_experts.err; echo "rc=$?" >> $O/bench_experts.err
timeout 200 python bench.py --env MultiTaskFetchArm8-v5 --rollout-batch-size 1024 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_arm8_1024.json 2> $O/bench_arm8_1024.err; echo "rc=$?" >> $O/bench_arm8_1024.err
for m in 0 4 8; do CURIOUS_XCD_MAP=$m timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --phases > $O/bench_xcd$m.json 2> $O/bench_xcd$m.err; done
tail -n 3 $O/pytest.txt; for f in $O/*.err; do echo $f; tail -n 2 $f; done; nproc
