#!/bin/bash
# the single-sample partition rules ("single_sample"): parity tests, then the latency regimes with the rule off / on (alternating on one box)
tag=${1:-r06ss}
out=gpurun_out/$tag
mkdir -p $out
timeout 1200 python -m pytest tests/test_conv_winograd.py tests/test_unet.py tests/test_pipeline.py tests/test_kernels.py -x -q -m gpu 2>&1 | tail -5 > $out/pytest.txt
for rep in 1 2; do
  for ks in 0 1; do
    echo "#### single_sample=$ks (rep $rep), default F(4x4) rule" >> $out/small.txt
    PROBE="64,1;32,1;32,16" PROBE_KSPLIT=$ks timeout 600 python tools/small_regime_probe.py 2>&1 | grep -v "amdgpu.ids" >> $out/small.txt
    echo "#### single_sample=$ks (rep $rep), F(4x4) rule 256 (AudioDiffusion's)" >> $out/small.txt
    PROBE="256,1" PROBE_RULE=256 PROBE_KSPLIT=$ks timeout 600 python tools/small_regime_probe.py 2>&1 | grep -v "amdgpu.ids" >> $out/small.txt
  done
done
cat $out/pytest.txt
grep "####\|==" $out/small.txt
