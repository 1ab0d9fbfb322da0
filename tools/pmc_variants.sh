#!/usr/bin/env bash
# On the GPU box: instruction / cycle counters of one kernel_times.py case for every prebuilt variant in csrc/variants/.
#   tools/pmc_variants.sh <case> [out_dir]
set -uo pipefail
cd "$(dirname "$0")/.."
CASE="${1:-C2}"; OUT="${2:-gpurun_out/pmc_variants}"
export TMPDIR=/tmp
mkdir -p "$OUT"
for lib in ray-tracer-challenge-rust_amd/csrc/variants/librtc_hip_*.so; do
  name="$(basename "$lib" .so)"; name="${name#librtc_hip_}"
  for pass in "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
              "write:WRITE_SIZE" \
              "cycles:SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    p="${pass%%:*}"; ctrs="${pass#*:}"
    # shellcheck disable=SC2086
    RTC_HIP_LIB="$PWD/$lib" timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/$name/$p" -- \
      python3 tools/kernel_times.py --cases "$CASE" --reps 3 > "$OUT/$name.$p.log" 2>&1 || echo "pass $name/$p failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, d)): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "render_kernel" in row["Kernel_Name"] and ", false, " in row["Kernel_Name"].split("render_kernel")[1][:40].replace("false, true>", "").replace("false, false>", "") + row["Kernel_Name"]:
                per[(row["Counter_Name"], row["Dispatch_Id"], row["Kernel_Name"])] += float(row["Counter_Value"])
        for (c, disp, k), v in per.items():
            import re
            if re.search(r"render_kernel<\w+, \w+, false", k): acc[c].append(v)
    print(d, {c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
