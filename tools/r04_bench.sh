R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r04k; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt
cut -c1-400 $O/bench_line.json; tail -4 $O/bench_err.txt
python - <<PY
import json
r=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print("value", r["value"], "roofline frac", r.get("roofline",{}).get("frac"), "fwd ms", r.get("roofline",{}).get("forward_ms"))
print("cpu", json.dumps(r.get("cpu_baseline"))[:300])
print("train", {k:v for k,v in r.get("train",{}).items() if k in ("ms_per_step","value","cpu_baseline","error")})
print("configs", {k:(v.get("ms_per_step"), v.get("cpu_baseline",{}).get("ms_per_step")) for k,v in r.get("configs",{}).items() if isinstance(v,dict)})
print("mel", r.get("mel",{}).get("cpu_baseline"), r.get("side_cpu_baselines"))
PY
