#!/usr/bin/env python3
"""EVERY pixel of BASELINE.json's five configurations at their full sizes, GPU (accelerated sweep, the kernel a caller gets) against
the CPU oracle (oracle/, cached-inverse mode, same arithmetic as the faithful mode): max |dcolor| per channel of the f32 and the f64
framebuffer, pixels beyond the 1e-5 bar, and the ray counts of both sides.  The test-suite compares a strided subsample of these
frames (tests/test_gpu_parity.py); this is the long form, run once per round and kept under profiles/.
   python tools/full_frame_parity.py [threads] [cases]        # cases: C1,C2,C3,C4,C5"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import torch  # noqa: E402,F401
import rtc_amd  # noqa: E402
from rtc_amd import scenes  # noqa: E402
import oracle_binding  # noqa: E402  (the checker: test infrastructure, never the product path)

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cases = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C1", "C2", "C3", "C4", "C5"]
TOL = 1e-5
bad = 0
for name in cases:
    cg, wg, L = scenes.CONFIGS[name](rtc_amd.api)
    co, wo, _ = scenes.CONFIGS[name](oracle_binding.api)
    img = cg.render(wg, L, 0, stats=True)
    plain = cg.render(wg, L, 0)
    assert (plain.rgb64.view(np.uint64) == img.rgb64.view(np.uint64)).all(), "counting and uncounted kernels differ"
    t0 = time.perf_counter()
    ref = np.zeros((cg.vsize, cg.hsize, 3))
    info = {"radiance_rays": 0, "shadow_rays": 0}
    bands = 16 if cg.hsize * cg.vsize > 4_000_000 else 2  # (a line of progress per band: the 4K glass frame takes minutes)
    for b in range(bands):
        y0, y1 = cg.vsize * b // bands, cg.vsize * (b + 1) // bands
        xs, ys = np.meshgrid(np.arange(cg.hsize), np.arange(y0, y1))
        part, pi = co.render_pixels(wo, xs.ravel(), ys.ravel(), L, nthreads=threads)
        ref[y0:y1] = part.reshape(y1 - y0, cg.hsize, 3)
        info["radiance_rays"] += pi["radiance_rays"]
        info["shadow_rays"] += pi["shadow_rays"]
        if bands > 2:
            print(f"  {name}: oracle rows {y0}..{y1 - 1} done after {time.perf_counter() - t0:.0f} s", flush=True)
    dt = time.perf_counter() - t0
    d32 = np.abs(img.rgb32.astype(np.float64) - ref)
    d64 = np.abs(img.rgb64 - ref)
    over = int((d32.max(axis=2) >= TOL).sum())
    same_rays = img.stats["radiance_rays"] == info["radiance_rays"] and img.stats["shadow_rays"] == info["shadow_rays"]
    bad += (over > 0) + (not same_rays)
    print(f"{name}: {cg.hsize}x{cg.vsize} L={L}  {cg.hsize * cg.vsize} pixels  max|d| f32 framebuffer {d32.max():.3e}  f64 framebuffer {d64.max():.3e}  "
          f"pixels >= {TOL:g}: {over}  rays GPU {img.stats['radiance_rays']}+{img.stats['shadow_rays']} oracle {info['radiance_rays']}+{info['shadow_rays']} "
          f"({'equal' if same_rays else 'DIFFERENT'})  brightest channel {np.abs(ref).max():.3f}  oracle {dt:.1f} s on {threads} threads", flush=True)
print("all pixels within the bar" if not bad else f"{bad} FAILED")
sys.exit(1 if bad else 0)
