R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/a8; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
A="--env MultiTaskFetchArm8-v5 --rollout-batch-size 1024 --no-cpu-baseline"
for s in 20 40 80; do python bench.py $A --steps $s --warmup 5 2>/dev/null | head -c 330 | tail -c 120; echo; done
python bench.py $A --steps 40 --warmup 45 2>/dev/null | head -c 330 | tail -c 120; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py $A --steps 40 --warmup 5 > /dev/null 2> $O/tr.log
cd $R; python tools/trace_gaps.py $O/tr --tail 0.3 > $O/gaps.txt 2>&1; cat $O/gaps.txt
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
