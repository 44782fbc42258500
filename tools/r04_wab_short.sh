# as tools/r04_wab.sh, one round only (when GPU minutes are short)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r04wabs}; mkdir -p $O
timeout 300 python -m pytest tests/test_conv_winograd.py -m gpu -x -q 2>&1 | tail -1
PROBE_SAVE=$O/new.pt timeout 200 python tools/forward_probe.py
ADM_LIB=$R/tools/libadm_hip_old.so PROBE_SAVE=$O/old.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/new.pt'); b=torch.load('$O/old.pt'); print('bit-identical forward:', torch.equal(a,b), float((a-b).abs().max()))"
timeout 200 python tools/forward_probe.py
