# round 5: what bounds conv_wino5_kernel — stage / role ablations and per-half cycle accounting (experiments build, one box)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w5abl}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py tests/test_conv_dispatch_random.py -m gpu -x -q 2>&1 | tail -3
L=$R/audio-diffusion_amd/audiodiffusion/libadm_hip_exp.so
for a in 0 7 56 32 16 48 55 64 1 2 4 128 8 119 0; do ADM_WINO5_ABL=$a ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/abl.txt; done
ADM_WINO5=0 ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/abl.txt
ADM_WINO5_PROF=1 PROBE_ONE=1 ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/abl.txt
for t in 0 1; do ADM_WINO5_TUNE=$t ADM_LIB=$L timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/abl.txt; done
