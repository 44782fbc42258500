# same-box A/B of two library builds on the bench forward (B = 32): tools/libadm_hip_old.so (built from the previous commit) vs the tree's
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04wab}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
PROBE_SAVE=$O/new.pt timeout 200 python tools/forward_probe.py
ADM_LIB=$R/tools/libadm_hip_old.so PROBE_SAVE=$O/old.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/new.pt'); b=torch.load('$O/old.pt'); print('bit-identical forward:', torch.equal(a,b), float((a-b).abs().max()))"
for i in 1 2; do
  timeout 200 python tools/forward_probe.py
  ADM_LIB=$R/tools/libadm_hip_old.so timeout 200 python tools/forward_probe.py
done
