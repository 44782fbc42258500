# Round-4 evidence run: gpurun -- 'bash tools/r04_final.sh [tag]'  (GPU suite with its wall time, smoke, bench line with the driver's flags,
# kernel trace of the bench command, PMC pass over one forward, kernel trace of the bf16 training step, PMC passes on the blocked-image
# kernels). The traced bench command carries --no-configs-leg: configs 2 / 4 launch the SAME Winograd kernel at B = 16 and would mix into
# the dominant kernel's average.
T=${1:-final}
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
bash tools/pmc_forward.sh $T/pmc r04 2>&1 | tail -25 > $O/pmc.txt
PROBE_MP=bf16 bash tools/profile_train_trace.sh $T/train > $O/train_trace.txt 2>&1
SKIP_TRACE=1 bash tools/r04_trace.sh $T/blk > $O/blk_pmc.txt 2>&1
PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > $O/small_regime.txt 2>&1
cat $O/pytest_gpu.txt; cat $O/pytest_gpu_durations.txt; grep '^==' $O/small_regime.txt; tail -1 $O/smoke.txt; tail -12 $O/blk_pmc.txt; cut -c1-300 $O/bench_line.json; tail -3 $O/bench_err.txt; tail -3 $O/train_trace.txt
