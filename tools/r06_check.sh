# round 6: the whole -m gpu suite + the forward / latency probes of the current build (one box)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06check}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.txt
for i in 1 2; do timeout 300 python tools/forward_probe.py 2>&1 | grep forward >> $O/forward.txt; done
timeout 300 python tools/layer_table_probe.py 2>&1 | grep -v amdgpu.ids > $O/layers_b32.txt
PROBE="64,1;32,16;256,1" timeout 300 python tools/small_regime_probe.py 2>&1 | grep "^==" > $O/small.txt
cat $O/pytest_gpu.txt $O/forward.txt $O/small.txt
