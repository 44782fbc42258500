# Round 4: in-call A/B of an environment-switched variant (one gpurun call, alternating processes on ONE box — box-to-box spread is
# +-0.5 ms on the training step, larger than most single changes).
#   gpurun -- 'bash tools/r04_ab.sh train ADM_BF16B_NARROW'      bf16 training step (B = 16) with VAR = 1 / 0
#   gpurun -- 'bash tools/r04_ab.sh small ADM_KSP_PIPE'          latency regimes (config 4: 32x32 B = 16; config 1: 64x64 B = 1)
# Switches that exist (all default 1 = the shipped behaviour): ADM_BF16B_NARROW (16- / 8-pixel rows on the blocked kernels), ADM_NT_STREAM /
# ADM_NT_CONV (streaming accesses of the elementwise passes / blocked epilogues), ADM_GN_FUSE_FINISH (split-K finish leaves the next
# GroupNorm's scale / shift), ADM_KSP_PIPE (register pipeline of the split generic kernel), ADM_KSP_1X1_SMALL (1x1 on <= 4x4 planes split).
R=$GRAFT_REPO_ROOT; cd $R; MODE=${1:-train}; VAR=${2:-ADM_BF16B_NARROW}; O=gpurun_out/r04_ab_$VAR; mkdir -p $O
for i in 1 2; do
  for v in 1 0; do
    if [ "$MODE" = train ]; then
      env $VAR=$v PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/${v}_$i.log 2>&1; echo "$VAR=$v run $i: $(grep 'train step' $O/${v}_$i.log)"
    else
      env $VAR=$v PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > $O/${v}_$i.log 2>&1; echo "$VAR=$v run $i:"; grep "^==" $O/${v}_$i.log
    fi
  done
done
