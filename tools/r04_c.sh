R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q -k "blocked or level3 or bf16_training_step" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
ADM_BF16B_PROF=1 PROBE_SHAPES="128,128,256;256,128,256;256,256,64" timeout 200 python tools/bf16b_probe.py > $O/prof.log 2>&1; grep "bf16b prof" $O/prof.log | sort | uniq -c | sort -rn | head -12
for L in 3; do
  ADM_BF16_LEVEL=$L PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_l$L.log 2>&1; echo "level $L: $(grep 'train step' $O/step_l$L.log)"
done
ADM_BF16_LEVEL=3 PROBE_MP=bf16 bash $R/tools/profile_train_trace.sh r04d/trace > $O/trace_head.txt 2>&1
head -32 $O/trace/train_kernel_stats.txt | cut -c1-150
