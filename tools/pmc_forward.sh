# HBM traffic per kernel over ONE UNet forward of the bench workload (B = 32, 256x256): two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; kernel-trace only) over tools/pmc_forward.py (two forwards: weights are packed in the first; per-kernel totals are
# halved), written in the format bench.py's roofline.traffic reads.   gpurun -- 'bash tools/pmc_forward.sh <outdir> <round tag>'
OUT=${1:-pmc_forward}; TAG=${2:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/$CTR -- python $R/tools/pmc_forward.py > $O/$CTR.log 2>&1
done
cd $R && python - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/audio-diffusion_amd")
from audiodiffusion import _native
_native.load()
from bench import kernel_stamp          # the stamp bench.py checks before it quotes this pass
O = "$O"
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.Counter()
for c in tot:
    for f in glob.glob(f"{O}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                tot[c][r["Kernel_Name"]] += float(r["Counter_Value"])
                if c == "FETCH_SIZE":
                    cnt[r["Kernel_Name"]] += 1
per = {}
for k in cnt:
    if "pack" in k:            # weight packing runs in the first forward only
        continue
    n = cnt[k] / 2.0
    per[k[:70]] = {"launches_per_forward": n, "fetch_KB_raw": tot["FETCH_SIZE"][k] / 2, "fetch_GB_corrected_x2": round(tot["FETCH_SIZE"][k] / 2 * 2 * 1024 / 1e9, 3),
                   "write_GB": round(tot["WRITE_SIZE"][k] / 2 * 1024 / 1e9, 3)}
def group(pred):
    ks = [k for k in cnt if pred(k) and "pack" not in k]
    n = sum(cnt[k] for k in ks) / 2.0
    f = sum(tot["FETCH_SIZE"][k] for k in ks) / 2 * 2 * 1024      # x2: gfx950 FETCH_SIZE reports half of 16 B/lane streaming reads
    w = sum(tot["WRITE_SIZE"][k] for k in ks) / 2 * 1024
    return {"kernel": " | ".join(sorted({k[:40] for k in ks})), "launches_per_forward": int(n), "fetch_bytes_per_forward": f,
            "write_bytes_per_forward": w, "hbm_bytes_per_launch": (f + w) / max(n, 1)}
out = {"stamp": kernel_stamp(), "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over two eager UNet forwards, B=32, 256x256 (tools/pmc_forward.sh), halved; "
                   "FETCH_SIZE doubled (gfx950 reports 1/2 of 16 B/lane streaming reads; check: gn_stats_kernel reads exactly its inputs — 21.2 GB per forward "
                   "when every GroupNorm runs the read pass (ADM_GN_FOLD=0), the inputs of the remaining read passes otherwise). "
                   "Infinity-Cache hits are counted by this counter.",
       "by_variant": {"4316": group(lambda k: "conv_wino6" in k), "4315": group(lambda k: "conv_wino5" in k),
                      "4314": group(lambda k: "conv_wino4" in k), "4313": group(lambda k: "conv_wino3" in k)},
       "gn_stats_check_GB": group(lambda k: "gn_stats" in k)["fetch_bytes_per_forward"] / 1e9,
       "per_kernel": per}
out["by_variant"] = {k: v for k, v in out["by_variant"].items() if v["launches_per_forward"]}
json.dump(out, open(f"{O}/${TAG}_pmc_forward.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
