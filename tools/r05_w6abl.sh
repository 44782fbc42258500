# timing-only ablations of conv_wino6_kernel (wrong results): pieces removed at compile time (tools/libadm_w6_*.so)
R=$GRAFT_REPO_ROOT; cd $R
for v in ${W6V:-"" NOP NOPF NOPL NOPE NOPFL}; do
  L=""; [ "$v" != "base" ] && [ -n "$v" ] && L=$R/tools/libadm_w6_$v.so
  echo "== variant [$v]"; ADM_LIB=$L ADM_WINO6=1 PROBE_ONE=1 timeout 120 python tools/wino5_abl_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
done
