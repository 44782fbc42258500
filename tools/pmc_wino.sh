# PMC passes (each its own rocprofv3 run, kernel-trace only beside --pmc) over tools/pmc_probe_wino.py (3 launches of the 128 -> 128 @256x256, B = 32
# Winograd convolution, GroupNorm + SiLU on load): issue / wait accounting, LDS activity and conflicts, instruction mix and HBM bytes of the
# kernel that ran (ADM_WINO6 / ADM_WINO5 select it).   gpurun -- 'bash tools/pmc_wino.sh <outdir>'
OUT=${1:-pmc_wino}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  WINO_MODE=4 timeout 90 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/pmc_probe_wino.py > $O/p$i.log 2>&1
done
python - <<PY | tee $O/summary.txt
import csv,glob,collections
agg=collections.defaultdict(float); dur=[]; names=set()
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); names.add(r['Kernel_Name'][:60])
for f in glob.glob("$O/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
n = max(1, len(dur))
print('kernel', sorted(names), 'launches', len(dur), 'us', ['%.0f'%d for d in dur])
for k in sorted(agg): print('%-28s %.6g  (per launch %.6g)'%(k, agg[k], agg[k]/n))
g = lambda k: agg.get(k, 0.0)
if g('SQ_LDS_IDX_ACTIVE'): print('SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f' % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
if g('SQ_WAVE_CYCLES'): print('SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f   SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f' % (g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')))
if g('SQ_INSTS_MFMA'): print('non-MFMA VALU per MFMA = %.2f   matrix pipe busy = %.3f (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))' % ((g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'), g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * g('GRBM_GUI_ACTIVE') / 8.0) if g('GRBM_GUI_ACTIVE') else 0))
print('HBM: FETCH_SIZE x2 (gfx950) %.3f GB + WRITE_SIZE %.3f GB per launch' % (g('FETCH_SIZE') / n * 2 * 1024 / 1e9, g('WRITE_SIZE') / n * 1024 / 1e9))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
