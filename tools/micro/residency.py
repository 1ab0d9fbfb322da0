#!/usr/bin/env python3
"""Diagnostics: resident recorded workgroups per XCD at a few instants of one launch (wave_timeline dump)."""
import sys
import numpy as np
d = dict(np.load(sys.argv[1]))
s, e, x = d["s_us"], d["e_us"], d["xcc"]
ts = [1, 2, 4, 8, 16, 24, 32, 40, 48, 56]
print("recorded workgroups", len(s), "span %.1f" % e.max())
print("t us      " + " ".join("%5d" % t for t in ts))
for xc in range(8):
    m = x == xc
    print("xcd %d    " % xc + " ".join("%5d" % int(((s[m] <= t) & (e[m] > t)).sum()) for t in ts))
print("all      " + " ".join("%5d" % int(((s <= t) & (e > t)).sum()) for t in ts))
