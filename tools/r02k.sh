R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --durations=8 2>&1 | tail -40 > $O/pytest_all.txt; tail -14 $O/pytest_all.txt
for MP in bf16 fp16; do timeout 200 python bench.py --mode train --mixed-precision $MP --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/train.txt; done
