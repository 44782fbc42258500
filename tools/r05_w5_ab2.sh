# round 5: v5 after a change — parity on the GPU, the two probe layers (product build, v5 / v4 alternating), cycle accounting, whole forward
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w5ab2}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py -m gpu -x -q 2>&1 | tail -2
L=$R/audio-diffusion_amd/audiodiffusion/libadm_hip_exp.so
for v in 1 0 1 0; do ADM_WINO5=$v timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; done
ADM_WINO5_PROF=1 PROBE_ONE=1 ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/ab.txt
for a in ${W5_ABLS:-7 55 119}; do ADM_WINO5_ABL=$a ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; done
ADM_WINO5=0 PROBE_SAVE=$O/v4.pt timeout 200 python tools/forward_probe.py
ADM_WINO5=1 PROBE_SAVE=$O/v5.pt timeout 200 python tools/forward_probe.py
python -c "import torch; a=torch.load('$O/v4.pt'); b=torch.load('$O/v5.pt'); print('bit-identical forward v4 vs v5:', torch.equal(a,b), float((a-b).abs().max()))"
ADM_WINO5=0 timeout 200 python tools/forward_probe.py
ADM_WINO5=1 timeout 200 python tools/forward_probe.py
