# Round-6 evidence run (ONE at the end of the round): gpurun -- 'bash tools/r06_final.sh [tag]'
# GPU suite with its wall time, smoke, the bench line with the driver's flags, the kernel trace of the bench command (--no-configs-leg: configs 2 / 4
# launch the same Winograd kernels at B = 16 and would mix into the dominant kernel's average), the STAMPED PMC pass over one forward
# (roofline.traffic), the PMC passes on conv_wino6_kernel and on the 1x1 class, the latency regimes (with the single-sample front end's
# per-model rule beside the default), the per-launch table, the training-step trace.
T=${1:-r06final}
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/$T; mkdir -p $O
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu_full.txt 2>&1; tail -6 $O/pytest_gpu_full.txt | grep -E "passed|failed|real" > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
bash tools/pmc_forward.sh $T/pmc r06 2>&1 | tail -25 > $O/pmc.txt
cp $O/pmc/r06_pmc_forward.json profiles/r06_pmc_forward.json          # so that the bench line below quotes THIS build's traffic
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs-leg > $R/$O/trace.log 2>&1
DB=$(find $R/$O/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/$O/kernel_stats.txt 2>&1
find $R/$O -name "*.db" -delete
cd $R
bash tools/pmc_wino.sh $T/pmc_v6 > $O/pmc_v6.txt 2>&1
bash tools/pmc_conv.sh $T/pmc_1x1 1x1 conv_mfma > $O/pmc_1x1.txt 2>&1
timeout 300 python tools/layer_table_probe.py 2>&1 | grep -v amdgpu.ids > $O/layers_b32.txt
PROBE="32,16;64,1;256,1" timeout 400 python tools/small_regime_probe.py > $O/small_regime.txt 2>&1
PROBE_RULE=256 PROBE="256,1" timeout 300 python tools/small_regime_probe.py 2>&1 | grep "^==" > $O/single_sample_rule.txt
( PROBE_KSPLIT=1 PROBE_RULE=256 PROBE="256,1" timeout 300 python tools/small_regime_probe.py; PROBE_KSPLIT=1 PROBE="64,1;32,1;32,16" timeout 300 python tools/small_regime_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/single_sample_on.txt
PROBE_MP=bf16 bash tools/profile_train_trace.sh $T/train > $O/train_trace.txt 2>&1
cat $O/pytest_gpu.txt; grep '^==' $O/small_regime.txt; cat $O/single_sample_rule.txt; grep '^==' $O/single_sample_on.txt; tail -1 $O/smoke.txt; cut -c1-300 $O/bench_line.json; tail -3 $O/bench_err.txt; tail -3 $O/train_trace.txt; tail -8 $O/pmc_v6.txt
