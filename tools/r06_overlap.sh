# round 6: side-stream overlap of the resnets' 1x1 shortcut convolutions (ADM_SIDE_OVERLAP=0 / 1): correctness on the GPU, then the latency regimes
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06overlap}; mkdir -p $O
timeout 1200 python -m pytest tests/test_unet.py tests/test_pipeline.py tests/test_vae.py tests/test_full_size.py tests/test_unet_condition.py -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
for v in 0 1 0 1; do
  echo "== ADM_SIDE_OVERLAP=$v" >> $O/overlap.txt
  ADM_SIDE_OVERLAP=$v PROBE="32,16;64,1;256,1;256,4" timeout 600 python tools/small_regime_probe.py 2>&1 | grep "^==" >> $O/overlap.txt
done
for v in 0 1; do ADM_SIDE_OVERLAP=$v PROBE_RULE=256 PROBE="256,1" timeout 300 python tools/small_regime_probe.py 2>&1 | grep "^==" | sed "s/^/[rule 256, overlap $v] /" >> $O/overlap.txt; done
cat $O/pytest.txt $O/overlap.txt
