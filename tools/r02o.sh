R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels.py tests/test_conv_dispatch_random.py tests/test_unet.py tests/test_reference_train_pin.py -m gpu -q -p no:cacheprovider -rfEs -s 2>&1 | tail -8 | tee $O/pytest.txt
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-mel-leg > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['frac'], r['forward_ms']); print(json.dumps(r['forward_breakdown']['conv_by_variant'])); print(r['forward_breakdown']['conv_small'], r['forward_breakdown']['groupnorm_stats'])
PY
