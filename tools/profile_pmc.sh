#!/usr/bin/env bash
# rocprofv3 passes for the bench workload (run on the GPU box):
#   1 kernel trace + stats, then PMC passes in their own runs (never combined with tracing domains other than
#   --kernel-trace), FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits, MI355X_MICROARCH.md).
#   tools/profile_pmc.sh <out_dir> [bench args...]
set -uo pipefail
cd "$(dirname "$0")/.."
OUT="${1:-gpurun_out/pmc}"; shift || true
ARGS=("$@")
[[ ${#ARGS[@]} -eq 0 ]] && ARGS=(--steps 5 --warmup 1 --no-cpu-baseline --no-parity)
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, rocprofv3 options...
  local name="$1"; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py "${ARGS[@]}" \
    > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed (rc=$?)"
}
run trace --kernel-trace --stats
run pmc_insts --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE
# GRBM_GUI_ACTIVE rides in the SAME pass as the SQ cycle counters (GRBM has its own two slots): VALU-busy and the mean
# number of resident waves are quotients of counters of one run
run pmc_cycles --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*.csv" | head -40
