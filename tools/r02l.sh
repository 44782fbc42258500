R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_mel.py tests/test_independent.py tests/test_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
for OC in 1 2; do ADM_MEL_OCC=$OC timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt; done
