#!/usr/bin/env bash
# On the GPU box: time every prebuilt variant in csrc/variants/ (tools/build_variants.sh) with tools/kernel_times.py.
#   tools/time_variants.sh [kernel_times.py args...]
set -uo pipefail
cd "$(dirname "$0")/.."
for lib in ray-tracer-challenge-rust_amd/csrc/variants/librtc_hip_*.so; do
  name="$(basename "$lib" .so)"; name="${name#librtc_hip_}"
  echo "=== $name"
  RTC_HIP_LIB="$PWD/$lib" timeout -k 10 300 python tools/kernel_times.py "$@" 2>&1 | grep -v amdgpu.ids
done
