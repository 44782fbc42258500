R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04x}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_conv_dispatch_random.py tests/test_vae.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  for v in "1 256" "0 256" "1 0"; do
    set -- $v
    ADM_KSP_1X1_SMALL=$1 ADM_KSP_PF_S8_COUT=$2 PROBE="32,16;64,1" timeout 300 python tools/small_regime_probe.py > $O/small_$1_$2_$i.txt 2>&1; echo "1x1small=$1 s8cout=$2 run $i:"; grep "^==" $O/small_$1_$2_$i.txt
  done
done
