R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
timeout 500 python -m pytest tests/test_reference_train_pin.py -m gpu -q -p no:cacheprovider -rfEs -s 2>&1 | tail -25 | tee $O/pytest.txt
