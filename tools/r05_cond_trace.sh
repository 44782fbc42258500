# Kernel trace of the conditional UNet forward (64x64 latents, B = 16): gpurun -- 'bash tools/r05_cond_trace.sh <outdir>'
OUT=${1:-r05cond}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${OUT}
cd /tmp && export TMPDIR=/tmp
PROBE_CHECK=${PROBE_CHECK:-0} PROBE_B=16 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${OUT}/tr -o tr -- python $R/tools/cond_probe.py > $R/gpurun_out/${OUT}/tr.log 2>&1
DB=$(find $R/gpurun_out/${OUT}/tr -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/${OUT}/cond_kernel_stats.txt 2>&1
find $R/gpurun_out/${OUT} -name "*.db" -delete
head -24 $R/gpurun_out/${OUT}/cond_kernel_stats.txt | cut -c1-170; grep "conditional UNet\|parity" $R/gpurun_out/${OUT}/tr.log
