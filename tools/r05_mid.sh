# round 5, mid-round check: the new / changed GPU tests and a short bench.py run (all legs)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05mid}; mkdir -p $O
timeout 900 python -m pytest tests/test_full_size.py -m gpu -x -q -k "complete_ddim50 or bench_batch_rows" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_unet_training.py tests/test_conv_dispatch_random.py tests/test_rccl_one_rank.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_short.json 2> $O/bench_short.err; tail -c 600 $O/bench_short.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_short.json") if l.startswith("{")][-1])
print("value", d["value"], "frac", d["roofline"].get("frac"), "kernel", d["roofline"].get("kernel","")[:40])
print("train", {k:d["train"].get(k) for k in ("ms_per_step","allreduce_buckets","allreduce_buckets_overlapped","allreduce_overlapped_on_every_rank","one_rank_group","error")})
print("mel", d["mel"]["forward"], d["mel"]["inverse"])
print("configs", {k:(v.get("ms_per_step"), v.get("error")) for k,v in d["configs"].items()})
print("cond cpu", d["configs"].get("conditional",{}).get("cpu_baseline"))
PY
