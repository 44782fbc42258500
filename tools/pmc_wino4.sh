# PMC passes (each its own rocprofv3 run, kernel-trace only) over tools/pmc_probe_wino.py (3 launches of the 128->128 @256x256 B=32
# Winograd conv): issue / wait accounting of conv_wino4_kernel. gpurun -- 'bash tools/pmc_wino4.sh <outdir> [ADM_WINO_ABL]'
OUT=${1:-pmc_wino4}; ABL=${2:-0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  WINO_MODE=${WINO_MODE:-4} ADM_WINO_ABL=$ABL timeout 90 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/pmc_probe_wino.py > $O/p$i.log 2>&1
done
python - <<PY | tee $O/summary.txt
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.Counter(); dur=[]
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for f in glob.glob("$O/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('mode ${WINO_MODE:-4} ABL $ABL launches', len(dur), 'us', ['%.0f'%d for d in dur])
for k in sorted(agg): print('%-28s %.6g  (per launch %.6g)'%(k, agg[k], agg[k]/max(1,len(dur))))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
