R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01j
cd /tmp && export TMPDIR=/tmp
PROBE_CHECK=0 PROBE_B=16 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01j/tr -o tr -- python $R/tools/gpu_probe.py trainstep > $R/gpurun_out/r01j/tr.log 2>&1
DB=$(find $R/gpurun_out/r01j/tr -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/r01j/train_kernel_stats.txt 2>&1
find $R/gpurun_out/r01j -name "*.db" -delete
head -30 $R/gpurun_out/r01j/train_kernel_stats.txt | cut -c1-150; grep "train step" $R/gpurun_out/r01j/tr.log
