# Round 4: narrow-row tilings of the blocked kernels: parity on the GPU, then the bf16 training step with and without them (A/B in one call)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04n}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 600 python -m pytest tests/test_full_size.py -m gpu -x -q -k "bf16" -s > $O/pytest_full.log 2>&1; echo "full-size rc=$?"; grep "bf16 level" $O/pytest_full.log; tail -1 $O/pytest_full.log
for i in 1 2; do
  for nar in 1 0; do
    ADM_BF16B_NARROW=$nar PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_${nar}_$i.log 2>&1; echo "narrow=$nar run $i: $(grep 'train step' $O/step_${nar}_$i.log)"
  done
done
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench train leg ms/step', r['ms_per_step'], r['value'])"
