# round 6, first call: per-launch table of the B = 32 and B = 1 forwards at HEAD, and the latency regimes' captured loops
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06base}; mkdir -p $O
timeout 300 python tools/layer_table_probe.py 2>&1 | grep -v amdgpu.ids > $O/layers_b32.txt
PROBE_B=1 timeout 300 python tools/layer_table_probe.py 2>&1 | grep -v amdgpu.ids > $O/layers_b1.txt
PROBE="64,1;32,16;256,1" timeout 300 python tools/small_regime_probe.py 2>&1 | grep "^==" > $O/small.txt
tail -3 $O/small.txt; head -1 $O/layers_b32.txt $O/layers_b1.txt
