# First GPU command of the next round: run everything that was written after round 1's GPU budget was spent and has
# therefore only ever run on the emulator.   gpurun --timeout 400 -- 'bash tools/first_contact.sh r02a'
#   1. the hardware variants of the first-contact tests (transformer-block backward kernels, conditional-UNet gradients,
#      opt-in bf16 1x1 convolutions) — ADM_TEST_UNTIMED=1 un-skips them
#   2. bf16 training step, level 1 (measured: 114.8 ms at B = 16) against level 2 (1x1 convolutions on bf16 operands too)
#      ... and the persistent chunk-stream forward kernel (ADM_BF16_PERSIST=1), per layer shape and on the whole step
#   3. the conditional UNet at the reference configuration, 64x64 latents: timing + parity against the oracle
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-first_contact}; mkdir -p $O
cd $R
ADM_BF16_PERSIST=1 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 90 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" | sed "s/^/bf16 level 1 + persistent forward kernel: /" | tee -a $O/train_levels.txt
for V in 0 1 2; do ADM_BF16_PERSIST=$V ADM_BF16_WIDE=0 timeout 40 python tools/bf16_ab_probe.py 2>&1 | grep -v "^TOTAL" | sed "s/^/persist=$V /" | tee -a $O/persist_ab.txt; done
ADM_BF16_8W=1 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 90 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" | sed "s/^/bf16 level 1 + 8-wave forward kernel: /" | tee -a $O/train_levels.txt
ADM_BF16_8W=1 ADM_BF16_WIDE=0 timeout 40 python tools/bf16_ab_probe.py 2>&1 | grep -v "^TOTAL" | sed "s/^/8-wave /" | tee -a $O/persist_ab.txt
ADM_WGRAD_BF16_8W=1 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 90 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" | sed "s/^/bf16 level 1 + 8-wave weight gradient: /" | tee -a $O/train_levels.txt
for L in 1 2; do
  ADM_BF16_LEVEL=$L PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 90 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" | sed "s/^/bf16 level $L: /" | tee -a $O/train_levels.txt
done
ADM_BF16_LEVEL=2 ADM_BF16_PERSIST=2 ADM_WGRAD_BF16_8W=1 PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 90 python tools/gpu_probe.py trainstep 2>&1 | grep "train step" | sed "s/^/bf16 level 2 + persistent 8-wave forward + 8-wave weight gradient: /" | tee -a $O/train_levels.txt
ADM_TEST_UNTIMED=1 ADM_BF16_LEVEL=2 timeout 90 python -m pytest tests/test_unet_training.py -m gpu -q -k mixed_precision 2>&1 | tail -2 | tee $O/level2_parity.txt
timeout 120 python tools/cond_probe.py 2>&1 | tail -3 | tee $O/cond_probe.txt
timeout 90 python tools/bf16_blocked_probe.py 2>&1 | tail -5 | tee $O/blocked_probe.txt
