# round 6: Mel codec timings of the current build (forward at B = 256, inverse at B = 32; ADM_MEL_OCC = register allocation of the forward kernel)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06mel}; mkdir -p $O
for v in ${VARIANTS:-base}; do
  L=""; [ "$v" != "base" ] && L=$R/tools/variants/libadm_$v.so
  for occ in 2 1 2; do echo "[$v occ $occ]" >> $O/mel.txt; ADM_LIB=$L ADM_MEL_OCC=$occ timeout 200 python tools/mel_probe.py 2>&1 | grep "forward power\|inverse" >> $O/mel.txt; done
done
timeout 900 python -m pytest tests/test_mel.py tests/test_independent.py tests/test_thirdparty_pin.py tests/test_full_size.py -m gpu -x -q -k "mel or Mel or pin or independent" 2>&1 | tail -3 >> $O/mel.txt
cat $O/mel.txt
