#!/bin/bash
# Round 3: kernel traces (durations + gaps) of the single-rank and the forced one-rank RCCL cycle
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv -d $O/t_single -- $B > $O/single.json 2> $O/single.log
CURIOUS_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $O/t_rccl -- $B > $O/rccl.json 2> $O/rccl.log
CURIOUS_FORCE_DIST=1 CURIOUS_GRAPH_ALLREDUCE=0 rocprofv3 --kernel-trace --output-format csv -d $O/t_rccl_eager -- $B > $O/rccl_eager.json 2> $O/rccl_eager.log
cd $R
for t in t_single t_rccl t_rccl_eager; do python tools/trace_gaps.py $O/$t --tail 0.3 > $O/$t.txt 2>&1; done
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
CURIOUS_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl.json 2> $O/bench_rccl.err
CURIOUS_FORCE_DIST=1 CURIOUS_GRAPH_ALLREDUCE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_eager.json 2> $O/bench_rccl_eager.err
CURIOUS_FORCE_DIST=1 CURIOUS_ASYNC_STORE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_sync.json 2> $O/bench_rccl_sync.err
cat $O/t_rccl.txt
