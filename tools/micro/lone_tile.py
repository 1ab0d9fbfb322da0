#!/usr/bin/env python3
"""How long do the waves of ONE tile of the C3 frame live when nothing else runs on the machine?  A camera of 8 x 8 (or n x n tiles)
pixels that looks through the window of the 1920 x 1080 frame at tile (tx, ty): same pixel size, half_width / half_height moved by the
window's offset (the rays agree with the frame's to rounding).  Compare with the tile's wave life inside the full frame
(tools/wave_timeline.py).    python tools/micro/lone_tile.py [tx ty [tiles_w tiles_h]]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import torch  # noqa: E402
import rtc_amd  # noqa: E402
from rtc_amd import abi, scenes  # noqa: E402

tx, ty = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (126, 42)
tw, th = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 1)
cam, w, L = scenes.c3_teapot(rtc_amd.api)
sc = w.upload(0)
full = cam.desc()
d = abi.RtcCameraDesc()
d.hsize, d.vsize = 8 * tw, 8 * th
d.pixel_size = full.pixel_size
d.half_width = full.half_width - tx * 8 * full.pixel_size
d.half_height = full.half_height - ty * 8 * full.pixel_size
for i in range(16):
    d.transform_inverse[i] = full.transform_inverse[i]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
buf = torch.zeros((d.vsize, d.hsize, 3), dtype=torch.float32, device=dev)
big = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev)
opts = abi.RtcRenderOpts(L, 0, 0, 0, 0)
for name, desc, out in (("window", d, buf), ("full frame", full, big)):
    for _ in range(5):
        sc.render_device(desc, opts, out.data_ptr(), 0, stream.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        sc.render_device(desc, opts, out.data_ptr(), 0, stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{name:10s} {desc.hsize}x{desc.vsize}: kernel {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})", flush=True)
same = torch.equal(buf, big[ty * 8: ty * 8 + d.vsize, tx * 8: tx * 8 + d.hsize])
print("window == the frame's pixels:", same, " max|d|", float((buf - big[ty * 8: ty * 8 + d.vsize, tx * 8: tx * 8 + d.hsize]).abs().max()))
