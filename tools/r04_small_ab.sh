# Round 4: latency regime A/B (configs 4 and 1): fused split-K finish + GroupNorm, cout tile of the split generic kernel, parts of the pf split
R=$GRAFT_REPO_ROOT
O=${1:-r04t}
mkdir -p $R/gpurun_out/$O
cd $R
run() {  # tag, env...
  tag=$1; shift
  env "$@" PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > gpurun_out/$O/$tag.txt 2>&1
  echo "== $tag: $@"; grep "^==" gpurun_out/$O/$tag.txt
}
run base ADM_GN_FUSE_FINISH=0
run fuse ADM_GN_FUSE_FINISH=1
run fuse_bm64 ADM_GN_FUSE_FINISH=1 ADM_KSP_BM=64
run fuse_bm64_s8 ADM_GN_FUSE_FINISH=1 ADM_KSP_BM=64 ADM_KSP_PF_S=8
run fuse_s8 ADM_GN_FUSE_FINISH=1 ADM_KSP_PF_S=8
timeout 600 python -m pytest tests/test_unet.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
