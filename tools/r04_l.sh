R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r04m; mkdir -p $O
timeout 300 python -m pytest tests/test_backward.py tests/test_unet_training.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  PROBE_CHECK=0 PROBE_B=16 PROBE_MP=bf16 timeout 200 python tools/gpu_probe.py trainstep > $O/step_$i.log 2>&1; echo "run $i: $(grep 'train step' $O/step_$i.log)"
done
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench train leg ms/step', r['ms_per_step'], r['value'])"
