#!/bin/bash
# Register / scratch / LDS use of every kernel in the device code of a built libnaruto_hip.so (or: compile the sources device-only
# first with `tools/kernel_resources.sh --compile`, which does not touch the in-tree library).
#   tools/kernel_resources.sh [--compile] [pattern]
set -e
L=/opt/rocm/lib/llvm/bin
HERE=$(cd "$(dirname "$0")/.." && pwd)
TMP=${TMPDIR:-/tmp}/naruto_kres.$$
mkdir -p "$TMP"
if [ "$1" = "--compile" ]; then
  shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics --cuda-device-only -c "$HERE/naruto_amd/csrc/naruto_api.hip" -o "$TMP/fat.bin" $NARUTO_EXTRA_FLAGS
  $L/clang-offload-bundler --unbundle --type=o --input="$TMP/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
else
  $L/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "${NARUTO_HIP_LIB:-$HERE/naruto_amd/libnaruto_hip.so}"
  $L/clang-offload-bundler --unbundle --type=o --input="$TMP/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
fi
$L/llvm-readelf --notes "$TMP/dev.co" | grep -E "\.name:|vgpr_spill|sgpr_spill|private_segment_fixed|\.vgpr_count|\.agpr_count|group_segment_fixed" \
  | paste - - - - - - - | sed 's/[ \t][ \t]*/ /g; s/_ZN6naruto[0-9]*//' \
  | awk '{n=$0; sub(/.*\.name: /,"",n); sub(/ .*/,"",n); printf "%-60s", substr(n,1,60); for(i=1;i<=NF;i++) if ($i ~ /count:|size:/) printf " %s %s", $i, $(i+1); printf "\n"}' \
  | grep -E "${1:-.}"
rm -rf "$TMP"
