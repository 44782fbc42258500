# round 5: conv_wino6_kernel (F(4x4,3x3)) on hardware — parity, per-layer A/B against the F(2x2) kernels, whole forward
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w6}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py tests/test_conv_dispatch_random.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do ADM_WINO6=$v timeout 300 python tools/wino_ab_probe.py 4 2>&1 | grep -v amdgpu.ids | tee $O/layers_w6_$v.txt; done
ADM_WINO6=0 PROBE_SAVE=$O/f2.pt timeout 200 python tools/forward_probe.py
ADM_WINO6=1 PROBE_SAVE=$O/f4.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/f2.pt'); b=torch.load('$O/f4.pt'); print('forward F(4x4) vs F(2x2): max|d|', float((a-b).abs().max()), 'max|ref|', float(a.abs().max()))"
ADM_WINO6=0 timeout 200 python tools/forward_probe.py
ADM_WINO6=1 timeout 200 python tools/forward_probe.py
timeout 600 python -m pytest tests/test_full_size.py -m gpu -x -q -k "unet_256 or bench_batch_rows or config3_256 or config2_256" 2>&1 | tail -3
