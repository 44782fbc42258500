R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04r}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_bf16.py tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_$i.log 2>&1; echo "run $i: $(grep 'train step' $O/step_$i.log)"
done
PROBE_MP=bf16 bash tools/profile_train_trace.sh ${1:-r04r}/train > $O/trace.txt 2>&1; grep -E "pack_|train step" $O/trace.txt | cut -c1-160
