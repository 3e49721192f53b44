R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R && python -m curious_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
CURIOUS_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $O/t_rccl -- $B > $O/rccl.json 2> $O/rccl.log
cd $R
python tools/trace_gaps.py $O/t_rccl --tail 0.55 > $O/t_rccl_all.txt 2>&1
python - <<'PY'
import csv,glob,re,os
O=os.environ.get('GRAFT_REPO_ROOT')+'/gpurun_out/r3w'
rows=[]
for p in glob.glob(O+'/t_rccl/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(.*$','',r['Kernel_Name']).replace('void ','')[:40]))
rows.sort()
# find a window in the middle of the timed region: 60% into the trace, print 12 consecutive kernels with gaps
i=int(len(rows)*0.6)
prev=rows[i-1][1]
out=open(O+'/t_rccl_window.txt','w')
for s,e,n in rows[i:i+16]:
    out.write('%-42s dur %7.2f us  gap %7.2f us\n'%(n,(e-s)/1e3,(s-prev)/1e3)); prev=e
out.close()
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/t_rccl_window.txt; head -8 $O/t_rccl_all.txt
