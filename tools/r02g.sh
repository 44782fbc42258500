R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_longform.py tests/test_conv_winograd.py tests/test_conv_dispatch_random.py tests/test_mel.py tests/test_independent.py tests/test_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 | tee $O/pytest.txt
timeout 120 python -m pytest tests/test_full_size.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest.txt
timeout 200 python tools/wino_ab_probe.py 4 2>&1 | grep -v amdgpu.ids | tee $O/wino_ab.txt
ADM_MEL_FAST=0 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
ADM_MEL_OCC=1 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
ADM_MEL_OCC=2 timeout 60 python tools/mel_probe.py 2>&1 | grep FAST | tee -a $O/mel.txt
