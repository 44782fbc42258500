# Round-end evidence run: gpurun -- 'bash tools/r03_final.sh'  (GPU suite, smoke, bench line with the driver's flags, kernel trace of
# the bench command, PMC pass over one forward, kernel trace of the bf16 training step). The traced bench command carries
# --no-configs-leg: configs 2 / 4 launch the SAME Winograd kernel at B = 16 and would mix into the dominant kernel's average.
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs-leg > $R/$O/trace.log 2>&1
DB=$(find $R/$O/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/$O/kernel_stats.txt 2>&1
find $R/$O -name "*.db" -delete
cd $R
bash tools/pmc_forward.sh final/pmc r03 2>&1 | tail -25 > $O/pmc.txt
PROBE_MP=bf16 bash tools/profile_train_trace.sh final/train > $O/train_trace.txt 2>&1
tail -1 $O/pytest_gpu.txt; tail -1 $O/smoke.txt; cut -c1-200 $O/bench_line.json; tail -3 $O/train_trace.txt
