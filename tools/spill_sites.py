#!/usr/bin/env python3
"""Where does a kernel touch scratch?  Reads hipcc -S --cuda-device-only -gline-tables-only output and lists, per
kernel, the scratch_load / scratch_store instructions grouped by the source line (.loc) they belong to.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -gline-tables-only \
        -o /tmp/k.s ray-tracer-challenge-rust_amd/csrc/rtc_hip.hip
  python tools/spill_sites.py /tmp/k.s render_kernelILb1ELb1ELb0
"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
files = {}
cur = None
inside = False
sites = collections.Counter()
n_inst = 0
for line in open(path):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r"^_Z\w+:", line):
        inside = pat in line
        continue
    if not inside:
        continue
    if line.startswith(".Lfunc_end"):
        inside = False
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    s = line.strip()
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    n_inst += 1
    if s.startswith("scratch_"):
        sites[(cur, s.split()[0])] += 1
print(f"{pat}: {n_inst} instructions, {sum(sites.values())} scratch accesses")
for (loc, op), n in sorted(sites.items(), key=lambda kv: (kv[0][0] or ("", 0))):
    print(f"  {loc[0]}:{loc[1]:<5} {op:<22} x{n}")
