R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py tests/test_conv_winograd.py tests/test_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_$i.log 2>&1; echo "run $i: $(grep 'train step' $O/step_$i.log)"
done
timeout 500 python -m pytest tests/test_full_size.py -m gpu -x -q -s -k "bf16_gradients and 3 or bench_batch_rows" > $O/pytest_gate.log 2>&1; echo "gate rc=$?"; grep -E "bf16 level|passed|failed|Error|assert" $O/pytest_gate.log | head -20
PROBE_MP=bf16 bash $R/tools/profile_train_trace.sh r04j/trace > $O/trace_head.txt 2>&1
head -36 $O/trace/train_kernel_stats.txt | cut -c1-150
