#!/usr/bin/env python3
"""Kernel-time table for a set of workloads (device-resident scene, f32 framebuffer only, HIP events).

  python tools/kernel_times.py [--reps 10] [--cases C3,C3x,empty,...]
  RTC_HIP_LIB=/path/to/variant/librtc_hip.so python tools/kernel_times.py ...     # time another build

Case names: <scene>[@WxH][:L<limit>][x]   (x = brute-force exact sweep).  `empty` = a world with no objects.

With a diagnostics build (tools/build_variants.sh "timers2:-DRTC_REGION_TIMERS=2", RTC_HIP_LIB=.../librtc_hip_timers2.so) and
RTC_TIMERS_UNCOUNTED=1 the shares of a wave's cycles per kernel region are printed for the ORDINARY (non-counting) kernel; mode 2
adds its samples up in LDS and carries no memory traffic into the timed regions (mode 1's global atomics distort the shares).
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))

import numpy as np  # noqa: E402

import rtc_amd  # noqa: E402
from rtc_amd import abi, scenes  # noqa: E402


def build(case):
    exact = case.endswith("x")
    if exact:
        case = case[:-1]
    limit = None
    if ":L" in case:
        case, l = case.split(":L")
        limit = int(l)
    size = {}
    if "@" in case:
        case, wh = case.split("@")
        w, h = wh.split("x")
        size = dict(hsize=int(w), vsize=int(h))
    api = rtc_amd.api
    if case == "empty":
        cam = api.camera(size.get("hsize", 1920), size.get("vsize", 1080), 0.785)
        cam.set_transform(api.view_transform((0, 4, -12), (0, 0, 0), (0, 1, 0)))
        world, L = api.world((0, 6.9, -5)), 5
    elif case in ("C3away", "C3zoom", "C3miss", "C3flip"):
        # the teapot scene with the camera turned away (every ray misses the mesh's group box) / zoomed in (the mesh
        # fills the frame): splits a frame's cost into per-background-pixel and per-mesh-pixel parts
        cam, world, L = scenes.c3_teapot(api, **size)
        if case == "C3away":  # NB the reference's group test covers the whole line: the mirror image still passes it
            cam.set_transform(api.view_transform((0.0, 4.0, -12.0), (0.0, 8.0, -24.0), (0.0, 1.0, 0.0)))
        elif case == "C3flip":  # the same view upside down: the launch's top-down tile order meets the teapot base first, its top last
            cam.set_transform(api.view_transform((0.0, 4.0, -12.0), (0.0, 0.0, 0.0), (0.0, -1.0, 0.0)))
        elif case == "C3miss":  # looking sideways: no pixel's line meets the group box (pure background cost)
            cam.set_transform(api.view_transform((0.0, 4.0, -12.0), (100.0, 4.0, -12.0), (0.0, 1.0, 0.0)))
        else:
            cam.set_transform(api.view_transform((0.0, 1.2, -4.2), (0.2, 0.1, 0.0), (0.0, 1.0, 0.0)))
    else:
        cam, world, L = scenes.CONFIGS[case](api, **size)
    return cam, world, (L if limit is None else limit), exact


REGIONS = ["raygen", "radiance_sweep", "prepare", "shadow_sweep", "lighting_children", "store"]


def region_shares(sc):
    """Region timers of a -DRTC_REGION_TIMERS=1 build (wave-resident cycles per kernel part), {} otherwise."""
    import ctypes as C
    lib = rtc_amd.hip()
    if not hasattr(lib, "rtc_debug_counters"):
        return {}
    buf = (C.c_ulonglong * 32)()
    lib.rtc_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.rtc_debug_counters.restype = C.c_int
    if lib.rtc_debug_counters(sc._h, buf) != 0 or buf[12] == 0:
        return {}
    out = {name: buf[6 + i] / buf[12] for i, name in enumerate(REGIONS)}
    if buf[25]:
        # q_filter includes the flushes that a full queue triggers inside the filter loop: report it net of nothing, as measured
        out["_sweep"] = {"setup_to_bundle": buf[21] / buf[25], "flush_exact": buf[22] / buf[25], "filter_rounds": buf[23] / buf[25],
                         "group_prep": buf[24] / buf[25], "sweep_share_of_wave": buf[25] / buf[12]}
        if buf[27] or buf[29]:  # -DRTC_REGION_TIMERS=2
            out["_sweep"].update({"exact_record_arrival": buf[27] / buf[25], "exact_arithmetic": buf[28] / buf[25],
                                  "group_boxes_of_wave": buf[29] / buf[12], "object_ray_of_wave": buf[30] / buf[12],
                                  "runs_ending_above_groups": buf[31] / buf[25]})
    out["_rounds"] = {"cluster_groups": buf[13], "tri_filter_rounds": buf[14], "exact_rounds": buf[15], "run_sweeps": buf[16],
                      "no_bundle_axis": buf[17], "tiles": buf[18], "per_ray_cluster_tests": buf[19]}
    if buf[27] or buf[29]:  # -DRTC_REGION_TIMERS=2 counts the runs that ended above the cluster groups in slot 17
        out["_rounds"]["runs_without_group"] = out["_rounds"].pop("no_bundle_axis")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cases", default="empty,C3:L0,C3,C3x,C3@3840x2160,C2,C4,C5")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    rows = []
    for case in args.cases.split(","):
        cam, world, L, exact = build(case)
        sc = world.upload(0)
        desc = cam.desc()
        buf = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev)
        flags = abi.RTC_FLAG_EXACT_SWEEP if exact else 0
        sc.reset_counters()
        sc.render_device(desc, abi.RtcRenderOpts(L, 0, 0, 0, flags | abi.RTC_FLAG_COUNT_RAYS), buf.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        st = sc.stats()
        if os.environ.get("RTC_TIMERS_UNCOUNTED"):  # a -DRTC_REGION_TIMERS=2 build times the ordinary kernels
            sc.reset_counters()
            sc.render_device(desc, abi.RtcRenderOpts(L, 0, 0, 0, flags), buf.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
        regions = region_shares(sc)
        opts = abi.RtcRenderOpts(L, 0, 0, 0, flags)
        for _ in range(2):
            sc.render_device(desc, opts, buf.data_ptr(), 0, stream.cuda_stream)
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            sc.render_device(desc, opts, buf.data_ptr(), 0, stream.cuda_stream)
            b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        rays = st["radiance_rays"] + st["shadow_rays"]
        ms = float(np.median(ts))
        row = {"case": case, "ms_median": round(ms, 4), "ms_min": round(min(ts), 4), "rays": rays,
               "Mrays_s": round(rays / ms / 1e3, 1) if ms > 0 else 0, "exact_tests": st["triangle_tests"],
               "filter_tests": st["filter_tests"], "box_tests": st["box_tests"], "prim_tests": st["primitive_tests"]}
        rows.append(row)
        sc.close()
        if not args.json:
            print(f"{case:>16s}  {ms:9.4f} ms (min {min(ts):8.4f})  rays {rays:>10d}  {row['Mrays_s']:>9.1f} Mrays/s  "
                  f"exact {st['triangle_tests']:>10d} filter {st['filter_tests']:>11d} box {st['box_tests']:>11d} prim {st['primitive_tests']}",
                  flush=True)
            if regions:
                rounds = regions.pop("_rounds")
                sw = regions.pop("_sweep", None)
                print(" " * 18 + "wave-cycle shares: " + "  ".join(f"{k} {v:.1%}" for k, v in regions.items()), flush=True)
                if sw:
                    print(" " * 18 + "triangle-run sweeps: " + "  ".join(f"{k} {v:.1%}" for k, v in sw.items()), flush=True)
                print(" " * 18 + "wave-level rounds: " + "  ".join(f"{k} {v}" for k, v in rounds.items()), flush=True)
    if args.json:
        print(json.dumps(rows))


if __name__ == "__main__":
    main()
