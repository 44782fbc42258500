R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q -k "blocked or level3 or bf16_training_step" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
PROBE_SHAPES="128,128,256;256,128,256;256,256,64;512,512,32" timeout 200 python tools/bf16b_probe.py > $O/probe.log 2>&1; grep -v amdgpu.ids $O/probe.log | cut -c1-330
for W in 512 256 384; do
  ADM_WGRADB_WGS=$W ADM_BF16_LEVEL=3 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_w$W.log 2>&1; echo "wgs $W: $(grep 'train step' $O/step_w$W.log)"
done
