R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels.py tests/test_conv_dispatch_random.py tests/test_unet.py tests/test_full_size.py tests/test_vae.py tests/test_pipeline.py tests/test_backward.py tests/test_unet_training.py tests/test_unet_condition.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/pytest.txt
python tools/layer_probe.py > $O/layers.txt 2>&1
ADM_CONV_KSPLIT=0 python tools/layer_probe.py 2>&1 | grep -E "forward" > $O/nosplit.txt
PROBE_B=16 PROBE_MP=bf16 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" > $O/step.txt
cat $O/pytest.txt; grep -E "forward" $O/layers.txt; grep "var  2311" $O/layers.txt | head -4; cat $O/nosplit.txt $O/step.txt
