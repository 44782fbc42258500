R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${1:-round_check}
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/${1:-round_check}/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${1:-round_check}/smoke.txt 2>&1
python bench.py > gpurun_out/${1:-round_check}/bench_line.json 2> gpurun_out/${1:-round_check}/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${1:-round_check}/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${1:-round_check}/trace.log 2>&1
DB=$(find $R/gpurun_out/${1:-round_check}/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/${1:-round_check}/kernel_stats.txt 2>&1
find $R/gpurun_out/${1:-round_check} -name "*.db" -delete
tail -1 $R/gpurun_out/${1:-round_check}/pytest_gpu.txt; tail -1 $R/gpurun_out/${1:-round_check}/smoke.txt; cut -c1-200 $R/gpurun_out/${1:-round_check}/bench_line.json
