# On the GPU box: time the default build and every variant in csrc/variants on the same box.   tools/micro/ab.sh [cases] [reps]
set -u
CASES=${1:-C3,C3zoom,C4,C5,C2}
REPS=${2:-10}
echo "== default build"; python tools/kernel_times.py --reps $REPS --cases $CASES 2>&1 | grep -v amdgpu
for lib in ray-tracer-challenge-rust_amd/csrc/variants/librtc_hip_*.so; do
  [ -e "$lib" ] || continue
  echo "== $(basename $lib .so)"; RTC_HIP_LIB=$PWD/$lib python tools/kernel_times.py --reps $REPS --cases $CASES 2>&1 | grep -v amdgpu
done
echo "== default build again"; python tools/kernel_times.py --reps $REPS --cases $CASES 2>&1 | grep -v amdgpu
