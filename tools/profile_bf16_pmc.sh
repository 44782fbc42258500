# One PMC pass (own run, kernel-trace only) over tools/pmc_probe_bf16.py: per-kernel duration, MFMA busy, LDS activity.
# gpurun -- 'bash tools/profile_bf16_pmc.sh <outdir>'
OUT=${1:-bf16_pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${OUT}
cd /tmp && export TMPDIR=/tmp
timeout ${PROBE_TIMEOUT:-60} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/${OUT}/p -- python $R/tools/pmc_probe_bf16.py > $R/gpurun_out/${OUT}/p.log 2>&1
python - <<PY > $R/gpurun_out/${OUT}/summary.txt 2>&1
import csv,glob,collections
fs=glob.glob("$R/gpurun_out/${OUT}/p/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[k]+=1
kt=glob.glob("$R/gpurun_out/${OUT}/p/*/*kernel_trace.csv")
dur=collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    if 'bf16' not in k: continue
    g=v['GRBM_GUI_ACTIVE']/8
    d=dur.get(k,[])
    print(k, 'launches=%d'%cnt[k], 'us(min/avg)=%s'%(('%.0f/%.0f'%(min(d),sum(d)/len(d))) if d else 'n/a'),
          'cycles/launch/xcd=%.3g'%(g/max(cnt[k],1)), 'mfma_busy=%.3f lds_active=%.3f lds_conflict=%.3f valu_per_mfmaBusy32=%.2f vmem_rd=%.3g'%(
          v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*g), v['SQ_LDS_IDX_ACTIVE']/(256*g), v['SQ_LDS_BANK_CONFLICT']/(256*g),
          v['SQ_INSTS_VALU']/max(v['SQ_VALU_MFMA_BUSY_CYCLES']/32,1), v['SQ_INSTS_VMEM_RD']))
PY
cat $R/gpurun_out/${OUT}/summary.txt; tail -2 $R/gpurun_out/${OUT}/p.log | cut -c1-200
find $R/gpurun_out/${OUT} -name "*.db" -delete
