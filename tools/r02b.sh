# r02 GPU call 2: whole -m gpu suite (now incl. loop-level oracle parity at BASELINE sizes, independent-derivation tests, 1-rank RCCL),
# then the default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs --durations=15 2>&1 | tail -120 > $O/pytest_all.txt
tail -30 $O/pytest_all.txt
timeout 400 python bench.py --steps 4 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
