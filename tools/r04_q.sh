# Round 4: streaming (non-temporal) accesses in the elementwise passes of the bf16 training step: parity, then A/B in one call
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04q}; mkdir -p $O
timeout 600 python -m pytest tests/test_backward.py tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  for nt in 1 0; do
    ADM_NT_STREAM=$nt PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_${nt}_$i.log 2>&1; echo "nt=$nt run $i: $(grep 'train step' $O/step_${nt}_$i.log)"
  done
done
