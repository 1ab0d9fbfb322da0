#!/usr/bin/env bash
# Build librtc_hip.so variants with different -D tuning macros and time them (run on the GPU box).
#   tools/variants.sh "name1:-DRTC_CLUSTER_TRIS=8 -DRTC_TILE_CLUSTERS=16" "name2:..." -- <kernel_times.py args>
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/variants
mkdir -p "$OUT"
CSRC=ray-tracer-challenge-rust_amd/csrc
specs=()
while [[ $# -gt 0 && "$1" != "--" ]]; do specs+=("$1"); shift; done
[[ $# -gt 0 ]] && shift
for spec in "${specs[@]}"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  lib="$PWD/$OUT/librtc_hip_$name.so"
  # shellcheck disable=SC2086
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function $flags \
    -o "$lib" "$CSRC/rtc_hip.hip" 2> "$OUT/build_$name.log" || { echo "build failed: $name"; tail -5 "$OUT/build_$name.log"; continue; }
  echo "=== $name ($flags)"
  RTC_HIP_LIB="$lib" timeout -k 10 300 python tools/kernel_times.py "$@" 2>&1 | grep -v amdgpu.ids | tee "$OUT/times_$name.txt"
done
