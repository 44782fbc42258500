#!/usr/bin/env bash
# A/B builds: tools/mkvariant.sh <name> <file.hip> [extra hipcc flags]  ->  tools/variants/libadm_<name>.so
# = the product library with ONE translation unit replaced by <file.hip> (same base name as the unit it replaces). The product's objects
# (audio-diffusion_amd/csrc/obj) must be current: run csrc/build.sh first. Loaded by the probes through ADM_LIB=<path>.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"; name="$1"; src="$2"; shift 2
unit="$(basename "$src" .hip)"; out="$root/tools/variants"; mkdir -p "$out"
extra=""; [ "$unit" = "k_conv_wino" ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -I"$root/audio-diffusion_amd/csrc" -I"$root/include" "$@" -c "$src" -o "$out/$unit.$name.o"
objs=(); for o in "$root"/audio-diffusion_amd/csrc/obj/*.o; do [ "$(basename "$o" .o)" = "$unit" ] && objs+=("$out/$unit.$name.o") || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libadm_$name.so" "${objs[@]}"
rm -f "$out/$unit.$name.o"; echo "built $out/libadm_$name.so"
