# round 5: plane-size floor of conv_wino6_kernel — the latency regimes (configs 1 / 4, B = 1) and the B = 32 forward per floor
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w6floor}; mkdir -p $O
for f in 0 64 128 256; do
  echo "== ADM_WINO6=$f" | tee -a $O/floor.txt
  ADM_WINO6=$f PROBE="64,1;64,16;256,1" timeout 300 python tools/small_regime_probe.py 2>&1 | grep "^==" | tee -a $O/floor.txt
  ADM_WINO6=$f timeout 200 python tools/forward_probe.py 2>&1 | grep forward | tee -a $O/floor.txt
done
