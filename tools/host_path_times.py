#!/usr/bin/env python3
"""Wall time of the drop-in call (Camera::render -> host flatten -> rtc_render -> framebuffers in host memory) per
frame, first call (upload + acceleration build) and repeat calls (scene found resident).   python tools/host_path_times.py [C3]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import rtc_amd  # noqa: E402
from rtc_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cam, w, L = scenes.CONFIGS[name](rtc_amd.api)
t = time.perf_counter()
w.flat_info()
print(f"{name}: host flatten (with value-equality classes) {1e3 * (time.perf_counter() - t):.2f} ms")
for i in range(6):
    t = time.perf_counter()
    img = cam.render(w, L, 0)
    print(f"call {i}: f32 + f64 framebuffers, no stats: wall {1e3 * (time.perf_counter() - t):.2f} ms")
H, W = cam.vsize, cam.hsize
rgb64 = np.zeros((H, W, 3))
rgb32 = np.zeros((H, W, 3), dtype=np.float32)
for what, a32, a64 in (("f64 only", None, rtc_amd._dptr(rgb64)), ("f32 only", rgb32.ctypes.data_as(rtc_amd._fp), None)):
    for i in range(3):
        t = time.perf_counter()
        rc = rtc_amd._host.rtch_render(cam._h, w._h, L, 0, 0, a32, a64, None)
        assert rc == 0
        print(f"{what}, call {i}: wall {1e3 * (time.perf_counter() - t):.2f} ms")
img = cam.render(w, L, 0, stats=True)
print("with stats (counting kernel):", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in img.stats.items()})
