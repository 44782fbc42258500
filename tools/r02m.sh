R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_reference_pin.py tests/test_thirdparty_pin.py tests/test_dataset_builder.py tests/test_kernels.py -m gpu -q -p no:cacheprovider -rfEs 2>&1 | tail -15 | tee $O/pytest.txt
timeout 120 python tools/wino_ab_probe.py 4 2>&1 | grep -v "^mode 4 variant" | tail -4 | tee $O/fwd.txt
