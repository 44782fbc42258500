# PMC pass (own run, kernel-trace only) over one training step: MFMA busy / LDS activity of the backward kernels.
# gpurun -- 'bash tools/profile_train_pmc.sh <outdir>'
OUT=${1:-train_pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${OUT}
cd /tmp && export TMPDIR=/tmp
PROBE_CHECK=0 PROBE_B=16 PROBE_MP=${PROBE_MP:-no} timeout ${PROBE_TIMEOUT:-300} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${OUT}/p -- python $R/tools/gpu_probe.py trainstep > $R/gpurun_out/${OUT}/p.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/${OUT}/p/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:48]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'wgrad_pf' in k or 'wino3' in k or 'gn_bwd' in k:
        g=v['GRBM_GUI_ACTIVE']/8
        print(k, 'mfma_busy_frac=%.3f lds_active_frac=%.3f lds_conflict_frac=%.3f'%(v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*g), v['SQ_LDS_IDX_ACTIVE']/(256*g), v['SQ_LDS_BANK_CONFLICT']/(256*g)))
PY
