#!/usr/bin/env python3
"""Who is resident when: per-workgroup start / end stamps of one render launch (diagnostics build).

  tools/build_variants.sh "timeline:-DRTC_WAVE_TIMELINE=1"
  RTC_HIP_LIB=$PWD/ray-tracer-challenge-rust_amd/csrc/variants/librtc_hip_timeline.so python tools/wave_timeline.py --cases C3,C5

Every workgroup (= one wave in the accelerated kernels) of the launch stamps s_memrealtime (100 MHz) at its first and last
instruction.  Printed per case: the launch's span, the distribution of wave lifetimes, the number of resident waves over time
(20 bins), when the last wave was dispatched, and which tiles the last-finishing waves rendered.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))

import rtc_amd  # noqa: E402
from rtc_amd import abi  # noqa: E402
from kernel_times import build  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="C3,C3zoom,C4,C5")
    ap.add_argument("--bins", type=int, default=20)
    ap.add_argument("--dump", default="")
    args = ap.parse_args()
    lib = rtc_amd.hip()
    if not hasattr(lib, "rtc_debug_timeline"):
        sys.exit("not a -DRTC_WAVE_TIMELINE=1 build (set RTC_HIP_LIB)")
    lib.rtc_debug_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint32]
    lib.rtc_debug_timeline.restype = C.c_int
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    N, REC = 1 << 18, 16
    for case in args.cases.split(","):
        cam, world, L, exact = build(case)
        sc = world.upload(0)
        desc = cam.desc()
        buf = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev)
        opts = abi.RtcRenderOpts(L, 0, 0, 0, abi.RTC_FLAG_EXACT_SWEEP if exact else 0)
        raw = (C.c_ulonglong * (N * REC))()
        for _ in range(3):
            sc.render_device(desc, opts, buf.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        lib.rtc_debug_timeline(raw, N)  # clears
        sc.render_device(desc, opts, buf.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        if lib.rtc_debug_timeline(raw, N) != 0:
            sys.exit("rtc_debug_timeline failed")
        a = np.frombuffer(raw, dtype=np.uint64).reshape(N, REC)
        ok = a[:, 3] != 0
        idx = np.nonzero(ok)[0]
        t0 = a[ok, 0].astype(np.int64)
        t1 = a[ok, 1].astype(np.int64)
        xcc = (a[ok, 2] & 0xF).astype(np.int64)
        bx = (a[ok, 3] & 0xFFFF).astype(np.int64)
        by = ((a[ok, 3] >> 16) & 0xFFFF).astype(np.int64)
        extra = a[ok, 4:].astype(np.int64)  # -DRTC_REGION_TIMERS=2 builds: per-wave rounds and region cycles
        have_extra = bool(extra.any())
        NAMES = ["groups", "filt_rounds", "exact_rounds", "run_sweeps", "runs_no_group", "tiles", "ray_cluster_tests",
                 "cyc_radiance", "cyc_shadow", "cyc_exact", "cyc_filter", "cyc_groupprep"]
        base = t0.min()
        s_us = (t0 - base) / 100.0
        e_us = (t1 - base) / 100.0
        life = e_us - s_us
        span = e_us.max()
        print(f"=== {case}: {len(idx)} workgroups, launch span {span:.1f} us (first stamp to last stamp)")
        qs = [50, 90, 99, 99.9, 100]
        print("  lifetimes us: " + "  ".join(f"p{q}={np.percentile(life, q):.2f}" for q in qs) + f"  sum={life.sum():.0f}")
        long_ = life > 1.0
        print(f"  waves living > 1 us: {long_.sum()}  (their lifetime sum {life[long_].sum():.0f} us, mean {life[long_].mean():.2f})")
        edges = np.linspace(0, span, args.bins + 1)
        occ = []
        for i in range(args.bins):
            lo, hi = edges[i], edges[i + 1]
            ov = np.clip(np.minimum(e_us, hi) - np.maximum(s_us, lo), 0, None).sum() / (hi - lo)
            occ.append(ov)
        print("  resident waves per time bin: " + " ".join(f"{o:.0f}" for o in occ))
        print(f"  last dispatch at {s_us.max():.1f} us; last long wave (> 1 us) dispatched at {s_us[long_].max():.1f} us")
        order = np.argsort(-e_us)[:12]
        print("  last finishers (end us, life us, xcc, tile bx,by, dispatch index):")
        for o in order:
            print(f"    {e_us[o]:8.2f} {life[o]:8.2f}  xcc{xcc[o]}  ({bx[o]},{by[o]})  #{idx[o]}")
        longest = np.argsort(-life)[:12]
        print("  longest-lived (life us, start us, xcc, tile, dispatch index):")
        for o in longest:
            print(f"    {life[o]:8.2f} {s_us[o]:8.2f}  xcc{xcc[o]}  ({bx[o]},{by[o]})  #{idx[o]}" +
                  ("  " + " ".join(f"{n}={v}" for n, v in zip(NAMES, extra[o])) if have_extra else ""))
        if have_extra:
            mesh = life > 4.0  # waves that swept something (background waves leave within ~2 us and never zero their counters)
            print(f"  waves living > 4 us: {mesh.sum()}; means: " + " ".join(f"{n}={extra[mesh, k].mean():.1f}" for k, n in enumerate(NAMES)))
            # how lifetime relates to the rounds: least-squares life ~ a*filt + b*exact + c*groups + d
            X = np.stack([extra[mesh, 1], extra[mesh, 2], extra[mesh, 0], extra[mesh, 5], np.ones(mesh.sum())], axis=1).astype(float)
            coef, *_ = np.linalg.lstsq(X, life[mesh], rcond=None)
            print("  life_us ~ %.4f*filt_rounds + %.4f*exact_rounds + %.4f*groups + %.4f*tiles + %.2f" % tuple(coef))
        # per-XCD end of work
        print("  per-XCD: end of last wave us / sum of lifetimes us: " +
              "  ".join(f"x{k}:{e_us[xcc == k].max():.1f}/{life[xcc == k].sum():.0f}" for k in range(8) if (xcc == k).any()))
        if args.dump:
            np.savez_compressed(f"{args.dump}_{case}.npz", idx=idx, s_us=s_us, e_us=e_us, xcc=xcc, bx=bx, by=by)
        sc.close()


if __name__ == "__main__":
    main()
