#!/usr/bin/env bash
# On the GPU box: one rocprofv3 PMC pass with the given counters on one kernel_times.py case; prints the mean per timed dispatch.
#   tools/pmc_custom.sh <case> <out_dir> COUNTER...
set -uo pipefail
cd "$(dirname "$0")/.."
CASE="$1"; OUT="$2"; shift 2
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/run" -- python3 tools/kernel_times.py --cases "$CASE" --reps 3 > "$OUT/log.txt" 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "run", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if re.search(r"render_kernel<\w+, \w+, false", row["Kernel_Name"]):
            per[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    for (c, d), v in per.items():
        acc[c].append(v)
print({c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
