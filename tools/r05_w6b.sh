# round 5: conv_wino6_kernel after a change — parity, layers, forward; old build (tools/libadm_hip_old.so) beside it
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r05w6b}; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_winograd.py -m gpu -x -q -k wino6 2>&1 | tail -2
ADM_WINO6=1 timeout 300 python tools/wino_ab_probe.py 4 2>&1 | grep -v amdgpu.ids | grep "4316\|forward" | tee $O/layers.txt
ADM_LIB=$R/tools/libadm_hip_old.so timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/old: /'
timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/new: /'
ADM_LIB=$R/tools/libadm_hip_old.so timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/old: /'
timeout 200 python tools/forward_probe.py 2>&1 | grep forward | sed 's/^/new: /'
