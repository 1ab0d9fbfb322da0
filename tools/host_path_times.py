import sys, time, os
sys.path.insert(0, "ray-tracer-challenge-rust_amd")
import rtc_amd
from rtc_amd import scenes
cam, w, L = scenes.c3_teapot(rtc_amd.api)
for i in range(3):
    t = time.perf_counter(); img = cam.render(w, L, 0); dt = time.perf_counter() - t
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in img.stats.items()}, "wall_ms", round(dt * 1e3, 2))
