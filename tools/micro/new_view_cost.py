#!/usr/bin/env python3
"""What a NEW view costs the host: the tile table of a launch (estimates, coverage, order) is built per (camera, rows) and cached; a
camera that moves pays for it every frame.  Frames of the same view against frames of ever-changing views (the camera's height moves
by a hundredth per frame), wall clock per frame including the launch.   python tools/micro/new_view_cost.py [C3|C4|...]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import torch  # noqa: E402
import rtc_amd  # noqa: E402
from rtc_amd import abi, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
api = rtc_amd.api
cam, w, L = scenes.CONFIGS[name](api)
sc = w.upload(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
fb = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev)
o = abi.RtcRenderOpts(L, 0, 0, 0, 0)
descs = []
for k in range(40):
    c = api.camera(cam.hsize, cam.vsize, 0.785)
    c.set_transform(api.view_transform((0.0 if name == "C3" else 8.0, 4.0 + 0.01 * k if name == "C3" else 6.0 + 0.01 * k, -12.0 if name == "C3" else -8.0),
                                       (0.0, 0.0 if name == "C3" else 3.0, 0.0), (0.0, 1.0, 0.0)))
    descs.append(c.desc())
for d in descs[:3]:
    sc.render_device(d, o, fb.data_ptr(), 0, st.cuda_stream)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(40):
    sc.render_device(descs[0], o, fb.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
same = (time.perf_counter() - t) / 40
t = time.perf_counter()
for d in descs[3:]:
    sc.render_device(d, o, fb.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
new = (time.perf_counter() - t) / len(descs[3:])
print(f"{name}: same view {same * 1e3:.3f} ms per frame (launch + sync), a new view every frame {new * 1e3:.3f} ms: the table costs {1e3 * (new - same):.3f} ms per new view")
