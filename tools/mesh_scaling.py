#!/usr/bin/env python3
"""How the frame time grows with the triangle count of the mesh: the C3 scene (teapot, same camera, same light, L = 5) with every
triangle of the teapot cut into 4^k coplanar ones (edge midpoints), k = 0 .. K.  The picture's geometry is the same, so rays and hits
stay what they are; what grows is the acceleration structure (clusters of 8 triangles, tiles of 64 clusters, and above 64 tiles the
super-tiles of 64 tiles).  Prints flatten / upload time, kernel time per frame and, for the small levels, whether the accelerated and
the brute-force sweep agree bit for bit.   python tools/mesh_scaling.py [K] [hsize] [vsize]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import torch  # noqa: E402
import rtc_amd  # noqa: E402
from rtc_amd import abi, meshes  # noqa: E402
from rtc_amd._capi import Material, Pattern  # noqa: E402


def subdivide(v, f):
    """every triangle -> 4 (edge midpoints; shared edges share the midpoint vertex)"""
    f0 = f.astype(np.int64) - 1
    edges = np.concatenate([f0[:, [0, 1]], f0[:, [1, 2]], f0[:, [2, 0]]])
    key = np.sort(edges, axis=1)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    mid = (v[uniq[:, 0]] + v[uniq[:, 1]]) * 0.5
    n = len(f0)
    m = inv.reshape(3, n).T + len(v)  # m[:, 0] = midpoint of (a, b), 1: (b, c), 2: (c, a)
    a, b, c = f0[:, 0], f0[:, 1], f0[:, 2]
    nf = np.concatenate([np.stack([a, m[:, 0], m[:, 2]], 1), np.stack([m[:, 0], b, m[:, 1]], 1),
                         np.stack([m[:, 2], m[:, 1], c], 1), np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)])
    return np.concatenate([v, mid]), (nf + 1).astype(np.int64)


def obj_text(v, f):
    lines = [f"v {float(x)!r} {float(y)!r} {float(z)!r}" for x, y, z in v]
    lines += [f"f {int(a)} {int(b)} {int(c)}" for a, b, c in f]
    return "\n".join(lines) + "\n"


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    api = rtc_amd.api
    v, f = meshes.load("teapot")
    f = f[:, :3]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    ref = None
    for k in range(K + 1):
        if k:
            v, f = subdivide(v, f)
        t0 = time.perf_counter()
        text = obj_text(v, f)
        t1 = time.perf_counter()
        cam = api.camera(W, H, 0.785)
        cam.set_transform(api.view_transform((0.0, 4.0, -12.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
        w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))
        g = api.obj_parse(text)
        g.set_transform(api.translation(0.0, -1.5, 0.0))
        g.set_material(Material(pattern=Pattern("gradient", (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))))
        w.push(g)
        t2 = time.perf_counter()
        sc = w.upload(0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        desc = cam.desc()
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        opts = abi.RtcRenderOpts(5, 0, 0, 0, 0)
        for _ in range(3):
            sc.render_device(desc, opts, frame.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        n = 20
        t4 = time.perf_counter()
        for _ in range(n):
            sc.render_device(desc, opts, frame.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t4) / n * 1e3
        sc.reset_counters()
        sc.render_device(desc, abi.RtcRenderOpts(5, 0, 0, 0, abi.RTC_FLAG_COUNT_RAYS), frame.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        st = sc.stats()
        rays = st["radiance_rays"] + st["shadow_rays"]
        note = ""
        if ref is None:
            ref = frame.clone()
        else:
            note = f"  max|d| vs level 0 {float((frame - ref).abs().max()):.3g}"
        if len(f) <= 30000 and W * H <= 1920 * 1080:
            exact = torch.zeros_like(frame)
            sc.render_device(desc, abi.RtcRenderOpts(5, 0, 0, 0, abi.RTC_FLAG_EXACT_SWEEP), exact.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
            note += f"  accel == brute force: {bool(torch.equal(exact, frame))}"
        print(f"level {k}: {len(f):8d} triangles  obj text {t1 - t0:6.2f} s  parse+tree {t2 - t1:6.2f} s  flatten+upload {t3 - t2:6.2f} s  "
              f"{ms:8.4f} ms per frame  {rays / ms / 1e3:9.1f} Mrays/s  box {st['box_tests']:>10d} filter {st['filter_tests']:>10d} "
              f"exact {st['triangle_tests']:>9d}{note}", flush=True)
        sc.close()


if __name__ == "__main__":
    main()
