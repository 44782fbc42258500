#!/usr/bin/env bash
# Build the product library for gfx950 (default) or the CPU-emulation TEST library (emu).
#   build.sh          -> audio-diffusion_amd/audiodiffusion/libadm_hip.so   (hipcc, gfx950)
#   build.sh emu      -> tests/emu/libadm_emu.so                          (g++ -DADM_EMU; tests only)
# (the -DADM_EXPERIMENTS builds of rounds 1-5 — superseded Winograd generations, ablation / cycle-accounting instantiations — were retired in
#  round 6 with the code they built; A/B builds of ONE translation unit: tools/mkvariant.sh)
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
srcs=(c_api.hip k_sched.hip k_groupnorm.hip k_conv_mfma.hip k_conv_wino.hip k_conv_wino_f2.hip k_conv_wino_f4.hip k_conv_bf16.hip k_conv_bf16b.hip k_conv1x1_bf16.hip k_conv_small.hip k_attention.hip k_transformer.hip k_audio_encoder.hip k_temb.hip k_train.hip k_backward.hip k_conv_wgrad.hip c_api_train.hip k_vae.hip net_exec.hip unet_exec.hip vae_exec.hip)
[ -f "$here/k_mel.hip" ] && srcs+=(k_mel.hip)
cd "$here"
exp=""; suf=""
if [ "${1:-hip}" = "emu" ]; then
  out="$root/tests/emu/libadm_emu$suf.so"
  objs=()
  mkdir -p "$root/tests/emu/obj$suf"
  for s in "${srcs[@]}"; do
    o="$root/tests/emu/obj$suf/${s%.hip}.o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ adm_rt.h -nt "$o" ] || [ adm_kernels.h -nt "$o" ] || [ net_exec.h -nt "$o" ] || [ k_conv_wino.h -nt "$o" ] || [ "$root/tests/emu/hip_emu.h" -nt "$o" ] || [ "$root/include/adm.h" -nt "$o" ]; then
      g++ -O2 -g -std=c++17 -fPIC -DADM_EMU $exp -I"$root/tests/emu" -x c++ -c "$s" -o "$o" -Wall -Wno-unknown-pragmas -Wno-unused-variable -Wno-unused-function -Wno-sign-compare -Wno-psabi &
    fi
    objs+=("$o")
  done
  wait
  relink=0
  for o in "${objs[@]}"; do if [ ! -f "$out" ] || [ "$o" -nt "$out" ]; then relink=1; fi; done
  if [ "$relink" = 1 ]; then   # link to a temporary name and rename: a concurrent loader never sees a half-written file
    g++ -shared -o "$out.tmp.$$" "${objs[@]}" -lpthread
    mv -f "$out.tmp.$$" "$out"
  fi
  echo "built $out"
else
  out="$root/audio-diffusion_amd/audiodiffusion/libadm_hip$suf.so"
  mkdir -p "$here/obj$suf"
  objs=()
  for s in "${srcs[@]}"; do
    o="$here/obj$suf/${s%.hip}.o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ adm_rt.h -nt "$o" ] || [ adm_kernels.h -nt "$o" ] || [ net_exec.h -nt "$o" ] || [ k_conv_wino.h -nt "$o" ] || [ "$root/include/adm.h" -nt "$o" ]; then
      # k_conv_wino*.hip: no SLP vectorisation — hipcc packs the scalar adds of the inverse transform into v_pk_add_f32 fed by ~2 v_mov each
      # (1813 -> 1333 VALU instructions in the kernel), and on gfx950 the fp32 MFMAs run on the same FMA lanes as the VALU: nothing is hidden
      extra=""; case "$s" in k_conv_wino*) extra="-fno-slp-vectorize";; esac
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $exp $extra -c "$s" -o "$o" &
    fi
    objs+=("$o")
  done
  wait
  relink=0
  for o in "${objs[@]}"; do if [ ! -f "$out" ] || [ "$o" -nt "$out" ]; then relink=1; fi; done
  if [ "$relink" = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out.tmp.$$" "${objs[@]}"
    mv -f "$out.tmp.$$" "$out"
  fi
  echo "built $out"
fi
