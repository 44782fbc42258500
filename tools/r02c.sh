R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_winograd.py tests/test_mel.py tests/test_backward.py tests/test_training.py tests/test_rccl_one_rank.py tests/test_golden.py tests/test_independent.py -m gpu -q -p no:cacheprovider -rfEs 2>&1 | tail -60 > $O/pytest.txt
tail -8 $O/pytest.txt
timeout 300 python tools/wino_ab_probe.py 3 4 2>&1 | grep -v amdgpu.ids | tee $O/wino_ab.txt
for M in 3 4; do WINO_MODE=$M ADM_WINO_PROF=1 timeout 60 python tools/pmc_probe_wino.py 2>&1 | grep -v amdgpu.ids | tail -4 | sed "s/^/mode $M: /" | tee -a $O/wino_prof.txt; done
