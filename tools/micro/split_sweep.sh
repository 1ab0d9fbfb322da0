# On the GPU box: sweep the thresholds above which the tile order splits a tile into halves / quarters (strip_table)
set -u
for cfg in "1e9 1e9" "0.7 1.1" "1e9 1.1" "1e9 0.9" "0.7 0.9" "0.6 0.9" "0.5 0.8" "0.9 1.2"; do
  set -- $cfg
  echo "== halves above $1, quarters above $2 of the frame estimate"
  RTC_SPLIT_ABOVE=$1 RTC_SPLIT4_ABOVE=$2 python tools/kernel_times.py --reps 10 --cases ${CASES:-C3,C3flip,C4,C3@3840x2160} 2>&1 | grep -v amdgpu | cut -c1-75
done
