#!/usr/bin/env bash
# On the GPU box: the round's rocprofv3 evidence for one workload: kernel trace + stats and the PMC passes of
# tools/profile_pmc.sh on `python3 bench.py --workload <W>`, summarised by tools/summarize_pmc.py.
#   tools/profile_round.sh <workload> <out_prefix> [steps]
set -uo pipefail
cd "$(dirname "$0")/.."
W="${1:-C3}"; OUT="${2:-gpurun_out/pmc_$W}"; STEPS="${3:-6}"
mkdir -p "$(dirname "$OUT")"
tools/profile_pmc.sh "$OUT" --workload "$W" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-parity --no-extra > "$OUT.log" 2>&1
python3 tools/summarize_pmc.py "$OUT" "${OUT}_summary" > /dev/null
cat "${OUT}_summary.txt"
