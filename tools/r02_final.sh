# Round-end evidence run: gpurun -- 'bash tools/r02_final.sh'  (GPU suite, smoke, bench line, kernel trace of the bench command,
# PMC pass over one forward, kernel trace of the bf16 training step)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_round_check.sh final
bash tools/pmc_forward.sh final/pmc r02 2>&1 | tail -25 > gpurun_out/final/pmc.txt
PROBE_MP=bf16 bash tools/profile_train_trace.sh final/train > gpurun_out/final/train_trace.txt 2>&1
tail -3 gpurun_out/final/train_trace.txt
