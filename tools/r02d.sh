R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
for A in 0 1 2 3 4 5; do ADM_WINO_ABL=$A timeout 60 python tools/wino_abl_probe.py 2>&1 | grep ABL | tee -a $O/abl.txt; done
WINO_MODE=3 timeout 60 python tools/wino_abl_probe.py 2>&1 | grep ABL | sed 's/^/mode3 /' | tee -a $O/abl.txt
