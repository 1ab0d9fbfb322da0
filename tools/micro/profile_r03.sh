# On the GPU box: the round's rocprofv3 evidence (kernel trace + stats, PMC passes) for C3 and C5, and the bench lines.
#   bash tools/micro/profile_r03.sh <out_dir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r03/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
tools/profile_round.sh C3 "$OUT/pmc_C3" > "$OUT/profile_c3.log" 2>&1
tools/profile_round.sh C5 "$OUT/pmc_C5" 4 > "$OUT/profile_c5.log" 2>&1
python bench.py --workload C5 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python tools/kernel_times.py --reps 10 --cases C3,C3flip,C3zoom,C3@3840x2160,C2,C4,C5,hexagon,table,C3x 2>&1 | grep -v amdgpu > "$OUT/times.txt"
find "$OUT" -name "*kernel_stats.csv" | head
tail -3 "$OUT/pmc_C3_summary.txt"; cat "$OUT/times.txt"
