# round 6: F(4x4) on the 32x32 planes (ADM_WINO6=32) against the default layer rule — captured loops at B = 1 / 4 / 16 and the 64x64 / latent models
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06rule}; mkdir -p $O
for w in default 32; do
  if [ $w = default ]; then unset ADM_WINO6; else export ADM_WINO6=$w; fi
  echo "== ADM_WINO6=$w" >> $O/rule.txt
  PROBE="256,1;256,4;256,16;64,1;32,16" timeout 600 python tools/small_regime_probe.py 2>&1 | grep "^==" >> $O/rule.txt
done
cat $O/rule.txt
