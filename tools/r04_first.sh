# Round 4, first GPU call: the blocked-image 16-bit kernels on hardware — parity, per-layer timings, the at-size bf16 gate, the step.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_bf16_blocked.py -m gpu -x -q > $O/pytest_blocked.log 2>&1; echo "blocked parity rc=$?"; tail -3 $O/pytest_blocked.log
timeout 300 python tools/bf16b_probe.py > $O/probe.log 2>&1; echo "probe rc=$?"; cat $O/probe.log | cut -c1-400
timeout 400 python -m pytest tests/test_unet_training.py -m gpu -x -q -k "level3 or bf16_training_step" > $O/pytest_train.log 2>&1; echo "train tests rc=$?"; tail -3 $O/pytest_train.log
timeout 500 python -m pytest tests/test_full_size.py -m gpu -x -q -s -k "bf16_gradients" > $O/pytest_gate.log 2>&1; echo "gate rc=$?"; grep -E "bf16 level|passed|failed|Error|assert" $O/pytest_gate.log | head -20
for L in 3 2; do
  ADM_BF16_LEVEL=$L PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_l$L.log 2>&1; echo "level $L: $(grep 'train step' $O/step_l$L.log)"
done
