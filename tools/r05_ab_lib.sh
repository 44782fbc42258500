# A/B of two builds on one box: tools/libadm_hip_old.so (the build before a change) vs the product library; whole forward, alternating
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05ablib}; mkdir -p $O
timeout 300 python -m pytest tests/test_conv_winograd.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2 3; do
  ADM_LIB=$R/tools/libadm_hip_old.so PROBE_SAVE=$O/old.pt timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/old: /'
  PROBE_SAVE=$O/new.pt timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/new: /'
done
python -c "import torch; a=torch.load('$O/new.pt'); b=torch.load('$O/old.pt'); print('bit-identical forward:', torch.equal(a,b), float((a-b).abs().max()))"
