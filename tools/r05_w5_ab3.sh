# round 5: the interleaved schedule (ADM_WINO5=5) against the two-halves schedule (1) and v4 (0), one box
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w5i}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py -m gpu -x -q 2>&1 | tail -2
for v in 5 1 0 5 1 0; do ADM_WINO5=$v timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; done
ADM_WINO5=0 PROBE_SAVE=$O/v4.pt timeout 200 python tools/forward_probe.py
ADM_WINO5=5 PROBE_SAVE=$O/v5.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/v4.pt'); b=torch.load('$O/v5.pt'); print('bit-identical forward v4 vs v5i:', torch.equal(a,b), float((a-b).abs().max()))"
ADM_WINO5=1 timeout 200 python tools/forward_probe.py
ADM_WINO5=5 timeout 200 python tools/forward_probe.py
