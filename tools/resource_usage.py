#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel in rtc_hip.hip (hipcc -Rpass-analysis=kernel-resource-usage).
   python tools/resource_usage.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "ray-tracer-challenge-rust_amd", "csrc", "rtc_hip.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/rtc_resource_usage.so", src] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel <REFR, ACCEL, COUNT, TRIS, POOLG>':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B':>9s} {'LDS B':>7s} {'waves/SIMD':>10s}")
for k, r in rows.items():
    print(f"{k:58s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('ScratchSize [bytes/lane]', 0):9d} "
          f"{r.get('LDS Size [bytes/block]', 0):7d} {r.get('Occupancy [waves/SIMD]', 0):10d}")
