# bf16 mixed-precision check on the GPU box: A/B of the two wave layouts of conv_bf16_kernel (event-timed), the bf16 parity
# tests, and a kernel trace of the full-size training step.  gpurun -- "bash tools/bf16_round_check.sh"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01s; mkdir -p $O
cd $R
ADM_BF16_WIDE=0 timeout 25 python tools/bf16_ab_probe.py > $O/ab0.txt 2>&1
ADM_BF16_WIDE=1 timeout 25 python tools/bf16_ab_probe.py > $O/ab1.txt 2>&1
grep -h "wide=" $O/ab0.txt $O/ab1.txt
T0=$(grep TOTAL $O/ab0.txt | awk '{print $3}'); T1=$(grep TOTAL $O/ab1.txt | awk '{print $3}')
W=$(python -c "print(1 if float('${T1:-1e9}') < float('${T0:-1e9}') else 0)")
echo "chosen wide=$W"
export ADM_BF16_WIDE=$W
timeout 30 python -m pytest tests/test_conv_bf16.py tests/test_backward.py -m gpu -q -k "bf16 or linear" 2>&1 | tail -2
PROBE_MP=bf16 PROBE_TIMEOUT=40 bash tools/profile_train_trace.sh r01s 2>&1 | head -8 | cut -c1-140
grep "train step" $O/tr.log
