R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_mel.py tests/test_independent.py tests/test_golden.py tests/test_dataset_builder.py tests/test_longform.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest.txt
timeout 100 python -m pytest tests/test_full_size.py -m gpu -q -p no:cacheprovider -k mel 2>&1 | tail -3 | tee -a $O/pytest.txt
ADM_MEL_FAST=0 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
ADM_MEL_OCC=1 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
ADM_MEL_OCC=2 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/trace -o mel -- python $R/tools/mel_probe.py > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB > $O/mel_kernel_stats.txt 2>&1; find $O -name "*.db" -delete
head -14 $O/mel_kernel_stats.txt | cut -c1-170
