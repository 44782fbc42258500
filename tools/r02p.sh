R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
timeout 500 python -m pytest tests/test_reference_pin.py -m gpu -q -p no:cacheprovider -rfEs -s -k longform 2>&1 | tail -15 | tee $O/pytest.txt
