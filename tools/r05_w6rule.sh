# round 5: the F(4x4) layer rule — plane floor 128 ("wino6" = 1) against ">= 64x64 and >= 32 workgroups per sample" ("wino6" = 3)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w6rule}; mkdir -p $O
for f in 1 3 1 3; do
  echo "== ADM_WINO6=$f" | tee -a $O/rule.txt
  ADM_WINO6=$f PROBE="64,1;256,1;256,4;256,16" timeout 400 python tools/small_regime_probe.py 2>&1 | grep "^==" | tee -a $O/rule.txt
  ADM_WINO6=$f timeout 200 python tools/forward_probe.py 2>&1 | grep forward | tee -a $O/rule.txt
done
