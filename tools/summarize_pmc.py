#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile_pmc.sh into one JSON + a short text table.

  python tools/summarize_pmc.py gpurun_out/pmc_r01 profiles/r01/pmc_c3

Per counter: mean over the dispatches of the render kernel that ran WITHOUT the counting flag (the timed
launches; the first launch of bench.py is the counted one and is excluded).  FETCH_SIZE is reported raw and
doubled (MI355X_MICROARCH.md: on gfx950 it reads exactly half of a wide coalesced streaming read); both it and
WRITE_SIZE are in KiB units in rocprofv3's output.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load_counters(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "render_kernel" not in row.get("Kernel_Name", ""):
                    continue
                out[(row["Counter_Name"], row["Dispatch_Id"], row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    summary = {}
    for p in sorted(os.listdir(src)):
        d = os.path.join(src, p)
        if not os.path.isdir(d):
            continue
        if p == "trace":
            for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "render_kernel" in row["Name"]:
                            summary.setdefault("kernel_stats", []).append(
                                {k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
            continue
        per = defaultdict(dict)  # counter -> dispatch -> value (sum over dims)
        for (name, disp, kern), vals in load_counters(d).items():
            per[name][(int(disp), kern)] = per[name].get((int(disp), kern), 0.0) + sum(vals)
        for name, byd in per.items():
            items = sorted(byd.items())
            # the counted launch uses the <..., true, ...> (COUNT) instantiation: keep only COUNT=false dispatches of the
            # kernel that dominates (most dispatches), i.e. the timed launches of the bench workload
            timed = [v for (disp, kern), v in items if re.search(r"render_kernel<\w+, \w+, false", kern)] or [v for _, v in items[1:]] or [v for _, v in items]
            entry = {"mean_per_dispatch": sum(timed) / len(timed), "dispatches": len(timed)}
            if name == "GRBM_GUI_ACTIVE":
                summary.setdefault("GRBM_GUI_ACTIVE_by_pass", {})[p] = entry
                if p != "pmc_cycles" and "GRBM_GUI_ACTIVE" in summary:
                    continue
            summary[name] = entry
    if "FETCH_SIZE" in summary:
        f = summary["FETCH_SIZE"]["mean_per_dispatch"]
        summary["hbm_read_bytes_per_launch"] = {"raw_fetch_size_x1024": f * 1024, "corrected_x2": f * 2048}
    if "WRITE_SIZE" in summary:
        summary["hbm_write_bytes_per_launch"] = summary["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    if "hbm_read_bytes_per_launch" in summary and "hbm_write_bytes_per_launch" in summary:
        summary["hbm_traffic_bytes_per_launch"] = summary["hbm_read_bytes_per_launch"]["corrected_x2"] + summary["hbm_write_bytes_per_launch"]
    g = lambda k: summary.get(k, {}).get("mean_per_dispatch")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        wc = g("SQ_WAVE_CYCLES")
        summary["derived"] = {
            "valu_active_share_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / wc,
            "scalar_active_share_of_wave_cycles": (g("SQ_ACTIVE_INST_SCA") or 0) / wc,
            "wait_inst_any_share": (g("SQ_WAIT_INST_ANY") or 0) / wc,
            "wait_any_share": (g("SQ_WAIT_ANY") or 0) / wc,
        }
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            summary["derived"]["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        grbm = summary.get("GRBM_GUI_ACTIVE_by_pass", {}).get("pmc_cycles", {}).get("mean_per_dispatch")
        if grbm:
            # SQ_* cycle counters tick in quad-cycles and are summed over the chip's 1024 SIMDs (256 CUs x 4);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md): both from the SAME pass
            cycles = grbm / 8.0
            summary["derived"]["kernel_cycles"] = cycles
            summary["derived"]["valu_busy"] = g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cycles
            summary["derived"]["mean_waves_per_simd"] = wc * 4.0 / 1024.0 / cycles
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        summary.setdefault("derived", {})["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    ks = [k for k in summary.get("kernel_stats", []) if re.search(r"render_kernel<\w+, \w+, false", k["Name"])]
    if ks and "derived" in summary:
        avg_ns = float(ks[0]["AverageNs"])
        summary["derived"]["kernel_us_traced"] = avg_ns / 1e3
        if "kernel_cycles" in summary["derived"]:
            summary["derived"]["effective_clock_ghz"] = summary["derived"]["kernel_cycles"] / avg_ns
        if g("SQ_INSTS_VALU") and "valu_lane_utilisation" in summary["derived"]:
            # executed vector work against the chip's issue peak: 256 CUs x 4 SIMDs x 32 lanes per clock at 2.4 GHz
            # (= the 157.3 TFLOP/s FP32 peak / 2 flop per FMA); an f64 instruction occupies its lanes for two passes and
            # counts once here, so this understates the busy share (valu_busy above is the time-based figure)
            lane_ops = g("SQ_INSTS_VALU") * 64.0 * summary["derived"]["valu_lane_utilisation"]
            ach = lane_ops / (avg_ns * 1e-9) / 1e12
            summary["valu_issue"] = {"achieved": round(ach, 3), "peak": 78.6, "unit": "T lane-op/s", "frac": round(ach / 78.6, 4),
                                     "what": "SQ_INSTS_VALU x 64 x lane utilisation / traced kernel time vs 256 CU x 4 SIMD x 32 lanes x 2.4 GHz"}
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + ".json", "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    with open(dst + ".txt", "w") as fh:
        for k in sorted(summary):
            fh.write(f"{k}: {json.dumps(summary[k])}\n")
    print(open(dst + ".txt").read())


if __name__ == "__main__":
    main()
