R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q -k "blocked or level3 or bf16_training_step" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
for F in 1 0; do
  ADM_GN_FOLD_TRAIN=$F ADM_BF16_LEVEL=3 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_f$F.log 2>&1; echo "fold $F: $(grep 'train step' $O/step_f$F.log)"
done
timeout 500 python -m pytest tests/test_full_size.py -m gpu -x -q -s -k "bf16_gradients and 3" > $O/pytest_gate.log 2>&1; echo "gate rc=$?"; grep -E "bf16 level|passed|failed|Error|assert" $O/pytest_gate.log | head -20
