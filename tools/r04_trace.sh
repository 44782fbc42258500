# Round 4: kernel trace of the level-3 bf16 training step + PMC passes on the blocked-image kernels (128->128 @256^2, B = 16)
R=$GRAFT_REPO_ROOT
O=${1:-r04b}
mkdir -p $R/gpurun_out/$O
if [ "${SKIP_TRACE:-0}" != "1" ]; then
ADM_BF16_LEVEL=3 PROBE_MP=bf16 bash $R/tools/profile_train_trace.sh $O/trace > $R/gpurun_out/$O/trace_head.txt 2>&1
head -45 $R/gpurun_out/$O/trace/train_kernel_stats.txt | cut -c1-170
fi
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/$O/pmc_$tag -- python $R/tools/pmc_probe_bf16b.py > $R/gpurun_out/$O/pmc_$tag.log 2>&1
done
python - <<PY > $R/gpurun_out/$O/pmc_summary.txt 2>&1
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
dur=collections.defaultdict(list)
for d in sorted(glob.glob("$R/gpurun_out/$O/pmc_*/")):
    fs=glob.glob(d+"*/*counter_collection.csv")
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'][:50]; c=r['Counter_Name']
        if c=='GRBM_GUI_ACTIVE': c=c+'@'+d.split('pmc_')[-1].strip('/')
        agg[k][c]+=float(r['Counter_Value']); cnt[k][c]+=1
    kt=glob.glob(d+"*/*kernel_trace.csv")
    if kt and 'MFMA' in d:
        for r in csv.DictReader(open(kt[0])):
            dur[r['Kernel_Name'][:50]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    if 'bf16b' not in k and 'wgradb' not in k and 'blk_apply' not in k: continue
    n=max(cnt[k].get('SQ_VALU_MFMA_BUSY_CYCLES',0),1)
    g=v.get('GRBM_GUI_ACTIVE@SQ_VALU_MFMA_BUSY_CYCLES',0)/8
    d=dur.get(k,[0])
    print(k, 'launches=%d us(min/avg)=%.0f/%.0f'%(n,min(d),sum(d)/len(d)))
    if g: print('   mfma_busy=%.3f lds_active=%.3f lds_conflict=%.3f valu/launch=%.3g vmem_rd/launch=%.3g'%(v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*g), v['SQ_LDS_IDX_ACTIVE']/(256*g), v['SQ_LDS_BANK_CONFLICT']/(256*g), v['SQ_INSTS_VALU']/n, v['SQ_INSTS_VMEM_RD']/n))
    if v.get('SQ_WAVE_CYCLES'): print('   wave_cycles/launch=%.4g wait_any=%.3f wait_inst_any=%.3f active_inst=%.3f lds_insts/launch=%.3g salu/launch=%.3g'%(v['SQ_WAVE_CYCLES']/n, v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_INSTS_LDS']/n, v['SQ_INSTS_SALU']/n))
    if v.get('FETCH_SIZE'): print('   FETCH_SIZE x2 per launch = %.4g GB (kB units x 2, MI355X_MICROARCH.md gfx950 correction)'%(v['FETCH_SIZE']/max(cnt[k]['FETCH_SIZE'],1)*1024*2/1e9))
    if v.get('WRITE_SIZE'): print('   WRITE_SIZE per launch = %.4g GB'%(v['WRITE_SIZE']/max(cnt[k]['WRITE_SIZE'],1)*1024/1e9))
PY
cat $R/gpurun_out/$O/pmc_summary.txt
find $R/gpurun_out/$O -name "*.db" -delete; find $R/gpurun_out/$O -name "*counter_collection.csv" -size +20M -delete
