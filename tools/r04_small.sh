# Round 4: the latency regime (config 4: 32x32, B = 16): GPU suite first, then a kernel trace of the eager forward + captured loop
R=$GRAFT_REPO_ROOT
O=${1:-r04s}
mkdir -p $R/gpurun_out/$O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$O/pytest.log 2>&1; tail -3 gpurun_out/$O/pytest.log
fi
PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > gpurun_out/$O/small.txt 2>&1
head -16 gpurun_out/$O/small.txt
cd /tmp && export TMPDIR=/tmp
PROBE="32,16" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$O/trace -- python $R/tools/small_regime_probe.py > $R/gpurun_out/$O/trace.log 2>&1
python - <<PY > $R/gpurun_out/$O/trace_summary.txt 2>&1
import csv,glob,collections
f=glob.glob("$R/gpurun_out/$O/trace/*/*kernel_trace.csv")[0]
rows=[(r['Kernel_Name'],int(r['Start_Timestamp']),int(r['End_Timestamp']),int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0, int(r.get('Workgroup_Size_X',0) or 0)) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:r[1])
# the last 186-ish launches before the end belong to the captured loop; print stats over all and one step in order
agg=collections.defaultdict(list)
for n,s,e,g,w in rows: agg[n[:90]].append((e-s)/1e3)
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:30]:
    print('%-92s n=%5d avg=%7.1f min=%7.1f total=%9.1f us (%.1f%%)'%(k,len(v),sum(v)/len(v),min(v),sum(v),100*sum(v)/tot))
# one step of the captured loop: find the last time_embedding launch and print the following kernels with gaps
idx=[i for i,r in enumerate(rows) if 'time_embedding' in r[0]]
if len(idx)>2:
    a,b=idx[-2],idx[-1]
    print('--- one step of the captured loop: %d launches, %.1f us wall'%(b-a,(rows[b][1]-rows[a][1])/1e3))
    prev=None
    for n,s,e,g,w in rows[a:b]:
        gap=(s-prev)/1e3 if prev else 0
        print('  %7.1f us  gap %6.1f  grid %7d wg %4d  %s'%((e-s)/1e3,gap,g,w,n[:100]))
        prev=e
PY
head -40 $R/gpurun_out/$O/trace_summary.txt
find $R/gpurun_out/$O -name "*.db" -delete
