# PMC passes (each its own rocprofv3 run, --kernel-trace only beside --pmc) over tools/pmc_probe_conv.py for one convolution class.
#   gpurun -- 'bash tools/pmc_conv.sh <outdir> <PROBE: 1x1|s2|8x8> <kernel name substring>'
OUT=${1:-pmc_conv}; export PROBE=${2:-1x1}; KSUB=${3:-conv_mfma}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/pmc_probe_conv.py > $O/p$i.log 2>&1
done
python - <<PY | tee $O/summary.txt
import csv,glob,collections
agg=collections.defaultdict(float); dur=[]; names=set()
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if '$KSUB' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); names.add(r['Kernel_Name'][:90])
for f in glob.glob("$O/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if '$KSUB' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('probe $PROBE kernels', sorted(names), 'launches', len(dur), 'us', ['%.0f'%d for d in dur])
for k in sorted(agg): print('%-28s %.6g  (per launch %.6g)'%(k, agg[k], agg[k]/max(1,len(dur))))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
