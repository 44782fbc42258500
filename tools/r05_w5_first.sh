# round 5, first hardware contact of conv_wino5_kernel: parity on the GPU, per-layer A/B against v4, whole-forward A/B (one box)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w5}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py tests/test_conv_dispatch_random.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do ADM_WINO5=$v timeout 300 python tools/wino_ab_probe.py 4 2>&1 | tee $O/layers_w5_$v.txt; done
ADM_WINO5=0 PROBE_SAVE=$O/v4.pt timeout 200 python tools/forward_probe.py
ADM_WINO5=1 PROBE_SAVE=$O/v5.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/v4.pt'); b=torch.load('$O/v5.pt'); print('bit-identical forward v4 vs v5:', torch.equal(a,b), float((a-b).abs().max()))"
ADM_WINO5=1 ADM_WINO5_TUNE=0 timeout 200 python tools/forward_probe.py
ADM_WINO5=0 timeout 200 python tools/forward_probe.py
ADM_WINO5=1 timeout 200 python tools/forward_probe.py
