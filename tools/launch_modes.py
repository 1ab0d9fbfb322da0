#!/usr/bin/env python3
"""How the per-frame time of back-to-back rtc_render_device launches depends on what surrounds them (C3, resident scene):
plain launches, with the two HIP events bench.py records per step, frames alternating between two streams (two frames in
flight), and a hipGraph of the launches.   python tools/launch_modes.py [workload] [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))
import torch  # noqa: E402
import rtc_amd  # noqa: E402
from rtc_amd import abi, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cam, w, L = scenes.CONFIGS[name](rtc_amd.api)
dev = torch.device("cuda", 0)
sc = w.upload(0)
desc = cam.desc()
opts = abi.RtcRenderOpts(L, 0, 0, 0, 0)
bufs = [torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev) for _ in range(2)]
s0 = torch.cuda.Stream()  # two explicit streams: the legacy default stream would serialise against the other one
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream()


def timed(fn, label):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(K):
        fn(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"{label:58s} {dt * 1e3:8.4f} ms per frame", flush=True)


timed(lambda k: sc.render_device(desc, opts, bufs[0].data_ptr(), 0, s0.cuda_stream), "one stream, no events")
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]


def with_events(k):
    ev[k % K][0].record(s0)
    sc.render_device(desc, opts, bufs[0].data_ptr(), 0, s0.cuda_stream)
    ev[k % K][1].record(s0)


timed(with_events, "one stream, two HIP events per frame (bench.py)")
print(f"{'   mean event-to-event kernel time':58s} {sum(a.elapsed_time(b) for a, b in ev) / K:8.4f} ms")
streams = [s0, s1]
timed(lambda k: sc.render_device(desc, opts, bufs[k & 1].data_ptr(), 0, streams[k & 1].cuda_stream),
      "two streams, two framebuffers (two frames in flight)")
bufs.append(torch.zeros_like(bufs[0]))
streams = [s0, s1, s2]
timed(lambda k: sc.render_device(desc, opts, bufs[k % 3].data_ptr(), 0, streams[k % 3].cuda_stream),
      "three streams, three framebuffers")
