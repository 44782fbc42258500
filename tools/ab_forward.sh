# A/B of library builds on ONE box: ab_forward.sh <outdir> <variant names...>  ("base" = the product library). Alternating, two rounds:
# B = 32 forward (best of 6 eager profiles), the per-launch table of the last round, and torch.equal of the forward outputs against base.
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2; do for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$R/tools/variants/libadm_$v.so
  W6=""; case "$v" in w6_*) L=""; W6=${v#w6_};; esac     # "w6_<n>": the product library under ADM_WINO6=<n>
  export ADM_WINO6=$W6; [ -z "$W6" ] && unset ADM_WINO6
  echo -n "[$v] " >> $O/forward.txt
  ADM_LIB=$L PROBE_SAVE=/tmp/out_$v.pt timeout 300 python tools/forward_probe.py 2>&1 | grep forward >> $O/forward.txt
done; done
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$R/tools/variants/libadm_$v.so
  W6=""; case "$v" in w6_*) L=""; W6=${v#w6_};; esac
  export ADM_WINO6=$W6; [ -z "$W6" ] && unset ADM_WINO6
  ADM_LIB=$L timeout 300 python tools/layer_table_probe.py 2>&1 | grep -v amdgpu.ids > $O/layers_$v.txt
  python - <<PY >> $O/forward.txt
import torch
a, b = torch.load("/tmp/out_base.pt"), torch.load("/tmp/out_$v.pt")
print("[$v] output vs base: equal", torch.equal(a, b), "max|d|", float((a - b).abs().max()), "max|ref|", float(a.abs().max()))
PY
done
cat $O/forward.txt
