# round 5: accuracy budget of F(4x4) and of the fast exp / reciprocal (tools/accuracy_probe.py) -> profiles/r05_accuracy.md
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/acc; mkdir -p $O
timeout 900 python tools/accuracy_probe.py product 2>&1 | grep -v amdgpu.ids | tee $O/product.txt
ADM_LIB=$R/tools/libadm_precise.so timeout 900 python tools/accuracy_probe.py precise 2>&1 | grep -v amdgpu.ids | tee $O/precise.txt
python tools/accuracy_probe.py compare product precise | tee $O/compare.txt
