R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_conv_winograd.py tests/test_full_size.py tests/test_unet.py tests/test_pipeline.py tests/test_vae.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest.txt
for F in 0 1; do ADM_GN_FOLD=$F timeout 100 python tools/wino_ab_probe.py 4 2>&1 | grep "UNet forward" | sed "s/^/GN_FOLD=$F /" | tee -a $O/fwd.txt; done
timeout 300 python bench.py --steps 2 --warmup 1 --no-train-leg > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r["forward_ms"], r["traffic"], json.dumps(r["forward_breakdown"]["groupnorm_stats"]), d["mel"])
PY
