R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_bf16_blocked.py tests/test_unet_training.py -m gpu -x -q -k "blocked or level3 or bf16_training_step" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
for F in 1 0 1 0; do
  ADM_GNB=$F PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_g$F.log 2>&1; echo "gnb $F: $(grep 'train step' $O/step_g$F.log)"
done
PROBE_MP=bf16 bash $R/tools/profile_train_trace.sh r04g/trace > $O/trace_head.txt 2>&1
head -40 $O/trace/train_kernel_stats.txt | cut -c1-150
