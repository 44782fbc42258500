R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_mel.py tests/test_golden.py tests/test_longform.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
timeout 100 python -m pytest tests/test_full_size.py -m gpu -q -p no:cacheprovider -k mel 2>&1 | tail -2 | tee -a $O/pytest.txt
timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee $O/mel2.txt
