# round 5, last call: the GPU suite and the bench line of the HEAD build
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05sanity}; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "steps", d["steps"], "frac", r.get("frac"), "traffic", r.get("traffic"), "alg bytes", r.get("algorithmic_bytes_per_launch"))
print("configs", {k:v.get("ms_per_step") for k,v in d["configs"].items()}, "train", d["train"].get("ms_per_step"))
PY
