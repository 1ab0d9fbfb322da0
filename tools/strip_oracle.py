#!/usr/bin/env python3
"""Diagnostics: the strip order a launch would take if it knew every wave's life (from a tools/wave_timeline.py --dump file).

  python tools/strip_oracle.py dump.npz tiles_x tiles_y out.bin [strip=16]

Writes n_strips uint32 strip numbers (row-major strip index = tile_row * strips_per_row + tile_col // strip), longest measured wave
first, for RTC_STRIP_ORDER_FILE of a -DRTC_WAVE_TIMELINE=1 build: the bound on what a better estimate in strip_table() can gain.
"""
import sys

import numpy as np

d = np.load(sys.argv[1])
tiles_x, tiles_y, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
strip = int(sys.argv[5]) if len(sys.argv) > 5 else 16
n_sx = (tiles_x + strip - 1) // strip
life = d["e_us"] - d["s_us"]
key = np.zeros(n_sx * tiles_y)
idx = d["by"] * n_sx + d["bx"] // strip
np.maximum.at(key, idx, life)
key[key < 4.0] = 0.0  # background strips keep their row-major order
order = np.argsort(-key, kind="stable").astype(np.uint32)
order.tofile(out)
print(f"{out}: {len(order)} strips, {int((key > 0).sum())} with a wave > 4 us, longest {key.max():.1f} us")
