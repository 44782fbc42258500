R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_reference_pin.py tests/test_pipeline.py tests/test_full_size.py -m gpu -q -p no:cacheprovider -rfEs --durations=6 2>&1 | tail -18 | tee $O/pytest.txt
