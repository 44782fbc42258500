# Round-5 evidence run (ONE at the end of the round): gpurun -- 'bash tools/r05_final.sh [tag]'
# GPU suite with its wall time, smoke, the bench line with the driver's flags, the kernel trace of the bench command (--no-configs-leg:
# configs 2 / 4 launch the same Winograd kernels at B = 16 and would mix into the dominant kernel's average), the PMC pass over one forward
# (roofline.traffic), the PMC passes on the Winograd kernels at HEAD (v6 and, for the same box, v5 and v4), their cycle accounting, the conditional trace.
T=${1:-r05final}
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/$T; mkdir -p $O
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu_full.txt 2>&1; tail -6 $O/pytest_gpu_full.txt | grep -E "passed|failed|real" > $O/pytest_gpu.txt; grep -E "s (call|setup)" $O/pytest_gpu_full.txt | head -12 > $O/pytest_gpu_durations.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs-leg > $R/$O/trace.log 2>&1
DB=$(find $R/$O/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/$O/kernel_stats.txt 2>&1
find $R/$O -name "*.db" -delete
cd $R
bash tools/pmc_forward.sh $T/pmc r05 2>&1 | tail -25 > $O/pmc.txt
bash tools/pmc_wino4.sh $T/pmc_v6 > $O/pmc_v6.txt 2>&1
ADM_WINO6=0 ADM_WINO5=1 bash tools/pmc_wino4.sh $T/pmc_v5 > $O/pmc_v5.txt 2>&1
ADM_WINO6=0 ADM_WINO5=0 bash tools/pmc_wino4.sh $T/pmc_v4 > $O/pmc_v4.txt 2>&1
# cycle accounting: conv_wino6_kernel (developer build -DW6X_PROF of the same source, waves 0 / 4), conv_wino5_kernel (experiments build)
PROBE_ONE=1 ADM_LIB=$R/tools/libadm_w6_PROF.so timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 > $O/wino6_cycles.txt
L=$R/audio-diffusion_amd/audiodiffusion/libadm_hip_exp.so
for a in 1000 999 7 55; do ADM_WINO6=0 ADM_WINO5_ABL=$a ADM_LIB=$L PROBE_ONE=1 timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tail -2; done > $O/wino5_cycles.txt 2>&1
# the three Winograd generations on the same box, twice: F(4x4) v6, F(2x2) v5, F(2x2) v4
for v in "1 1" "0 1" "0 0" "1 1" "0 1" "0 0"; do set -- $v; ADM_WINO6=$1 ADM_WINO5=$2 timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/WINO6=$1 /"; done > $O/wino_ab.txt 2>&1
for f in 0 1 128 256; do ADM_WINO6=$f timeout 200 python tools/forward_probe.py 2>&1 | grep forward; done > $O/forward_by_floor.txt 2>&1
PROBE_CHECK=1 bash tools/r05_cond_trace.sh $T/cond > $O/cond.txt 2>&1
PROBE_MP=bf16 bash tools/profile_train_trace.sh $T/train > $O/train_trace.txt 2>&1
PROBE="32,16;64,1;256,1" timeout 400 python tools/small_regime_probe.py > $O/small_regime.txt 2>&1
tools/microbench/mfma_share > $O/mfma_share.txt 2>&1
cat $O/pytest_gpu.txt; cat $O/pytest_gpu_durations.txt; grep '^==' $O/small_regime.txt; tail -1 $O/smoke.txt; cut -c1-300 $O/bench_line.json; tail -3 $O/bench_err.txt; tail -3 $O/train_trace.txt; tail -4 $O/cond.txt
