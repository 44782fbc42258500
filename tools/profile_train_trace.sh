# Kernel trace of one training step (B=16) on the GPU box: gpurun -- 'bash tools/profile_train_trace.sh <outdir>'
# writes gpurun_out/<outdir>/train_kernel_stats.txt (copy the summary you want judged into profiles/).
OUT=${1:-train_trace}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${OUT}
cd /tmp && export TMPDIR=/tmp
PROBE_CHECK=0 PROBE_B=16 PROBE_MP=${PROBE_MP:-no} timeout ${PROBE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${OUT}/tr -o tr -- python $R/tools/gpu_probe.py trainstep > $R/gpurun_out/${OUT}/tr.log 2>&1
DB=$(find $R/gpurun_out/${OUT}/tr -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/${OUT}/train_kernel_stats.txt 2>&1
find $R/gpurun_out/${OUT} -name "*.db" -delete
head -30 $R/gpurun_out/${OUT}/train_kernel_stats.txt | cut -c1-150; grep "train step" $R/gpurun_out/${OUT}/tr.log
