#!/usr/bin/env bash
# Build librtc_hip.so variants HERE (hipcc cross-compiles gfx950 in seconds) into csrc/variants/, which travels to the
# GPU box with the snapshot (*.so is git-ignored, not gpurun-ignored); time them there with tools/time_variants.sh.
#   tools/build_variants.sh "name1:-DRTC_CAND_CAP=12" "name2:-DRTC_MIN_WAVES=2 ..."
set -euo pipefail
cd "$(dirname "$0")/.."
CSRC=ray-tracer-challenge-rust_amd/csrc
mkdir -p "$CSRC/variants"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  [[ "$flags" == "$spec" ]] && flags=""
  # shellcheck disable=SC2086
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function $flags \
    -o "$CSRC/variants/librtc_hip_$name.so" "$CSRC/rtc_hip.hip" &
done
wait
ls -la "$CSRC/variants"
