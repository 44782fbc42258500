# Round 4: register-pipelined split-K generic kernel: parity, then A/B on the latency regimes
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04v}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_conv_dispatch_random.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  for v in 1 0; do
    ADM_KSP_PIPE=$v PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > $O/small_${v}_$i.txt 2>&1; echo "pipe=$v run $i:"; grep "^==" $O/small_${v}_$i.txt
  done
done
