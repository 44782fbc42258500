R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_longform.py tests/test_pipeline.py tests/test_conv_winograd.py tests/test_conv_dispatch_random.py tests/test_full_size.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.txt
bash tools/pmc_forward.sh r02f/pmc r02 2>&1 | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-mel-leg > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.txt 2>&1; find $O -name "*.db" -delete
head -12 $O/kernel_stats.txt | cut -c1-200; tail -c 600 $O/trace.log
cd $R; timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
for A in 6 7 8 9 10 11; do ADM_WINO_ABL=$A timeout 60 python tools/wino_abl_probe.py 2>&1 | grep ABL | tee -a $O/abl2.txt; done
