# On the GPU box: A/B of the tile order (old = RTC_STRIP_ORDER=0 build in csrc/variants, new = the default build) and of the priority thresholds
set -u
mkdir -p gpurun_out/r03
V=$PWD/ray-tracer-challenge-rust_amd/csrc/variants
CASES=${1:-C3,C3flip,C3zoom,C4,C5}
echo "== old order"; RTC_HIP_LIB=$V/librtc_hip_oldorder.so python tools/kernel_times.py --reps 10 --cases $CASES 2>&1 | grep -v amdgpu
echo "== new order, no priorities"; RTC_PRIO1_AT=100 RTC_PRIO2_AT=100 RTC_PRIO3_AT=100 python tools/kernel_times.py --reps 10 --cases $CASES 2>&1 | grep -v amdgpu
echo "== new order, priorities at 0.25 / 0.4 / 0.6 of the frame estimate"; python tools/kernel_times.py --reps 10 --cases $CASES 2>&1 | grep -v amdgpu
echo "== new order, priorities at 0.15 / 0.25 / 0.4"; RTC_PRIO1_AT=0.15 RTC_PRIO2_AT=0.25 RTC_PRIO3_AT=0.4 python tools/kernel_times.py --reps 10 --cases $CASES 2>&1 | grep -v amdgpu
echo "== new order, priority 3 only, at 0.4"; RTC_PRIO1_AT=100 RTC_PRIO2_AT=100 RTC_PRIO3_AT=0.4 python tools/kernel_times.py --reps 10 --cases $CASES 2>&1 | grep -v amdgpu
echo "== timeline, no priorities"; RTC_PRIO1_AT=100 RTC_PRIO2_AT=100 RTC_PRIO3_AT=100 RTC_HIP_LIB=$V/librtc_hip_timeline.so python tools/wave_timeline.py --cases C3,C4 2>&1 | grep -E "===|resident|lifetimes|per-XCD" 
echo "== timeline, default priorities"; RTC_HIP_LIB=$V/librtc_hip_timeline.so python tools/wave_timeline.py --cases C3,C4 > gpurun_out/r03/timeline_striporder.txt 2>&1
grep -E "===|resident|lifetimes|per-XCD" gpurun_out/r03/timeline_striporder.txt
