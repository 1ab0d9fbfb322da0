#!/usr/bin/env python3
"""Longer run of the random-scene fuzz (tests/test_gpu_parity.py::_random_scene): accelerated vs brute-force sweep,
bit for bit (counting and uncounted kernels), over many more seeds than the test-suite spends time on.   python tools/fuzz_soak.py [first] [count]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
import rtc_amd  # noqa: E402
import rtc_amd._capi  # noqa: E402
import test_gpu_parity as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
bad = 0
for seed in range(first, first + count):
    try:
        cam, w, L = T._random_scene(rtc_amd.api, seed)
    except rtc_amd._capi.NativeError as e:  # e.g. a random shear that is singular: the reference panics too
        print("seed", seed, "skipped:", e, flush=True)
        continue
    fast = cam.render(w, L, 0, stats=True)
    slow = cam.render(w, L, 0, exact_sweep=True, stats=True)
    plain = cam.render(w, L, 0)  # the uncounted kernel a caller gets (tile table, cleared tiles)
    same = (fast.rgb64.view(np.uint64) == slow.rgb64.view(np.uint64)) & (plain.rgb64.view(np.uint64) == slow.rgb64.view(np.uint64))
    if not same.all() or fast.stats["radiance_rays"] != slow.stats["radiance_rays"] or fast.stats["shadow_rays"] != slow.stats["shadow_rays"]:
        bad += 1
        print("MISMATCH seed", seed, int((~same).sum()), "values", flush=True)
    if (seed - first) % 250 == 249:
        print("seeds", first, "..", seed, "mismatching scenes:", bad, flush=True)
print("done:", count, "scenes,", bad, "mismatching")
sys.exit(1 if bad else 0)
