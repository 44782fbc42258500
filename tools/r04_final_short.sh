# Short end-of-round evidence run (GPU minutes nearly spent): the GPU tests that exercise the last change (Winograd producers) and the at-size
# parity tests, smoke, the bench line with the driver's flags, the kernel trace of the bench command. The other artefacts (PMC passes, training
# trace, latency-regime records) stay from tools/r04_final.sh's last full run — the code they measure did not change.
T=${1:-final_short}
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/$T; mkdir -p $O
( time python -m pytest tests/test_conv_winograd.py tests/test_full_size.py tests/test_pipeline.py tests/test_unet.py -m gpu -x -q ) > $O/pytest_gpu_subset.txt 2>&1; tail -4 $O/pytest_gpu_subset.txt | grep -E "passed|failed|real"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt
cut -c1-220 $O/bench_line.json; tail -3 $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs-leg > $R/$O/trace.log 2>&1
DB=$(find $R/$O/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/$O/kernel_stats.txt 2>&1
find $R/$O -name "*.db" -delete
grep conv_wino4 $R/$O/kernel_stats.txt | cut -c1-160
