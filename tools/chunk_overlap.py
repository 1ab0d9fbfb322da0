#!/usr/bin/env python3
"""One rank's share of a row-sharded frame on ONE device: how long do its chunk launches take when they are queued on one
stream (each chunk's long-lived waves drain before the next chunk starts) and when they alternate between streams (the tails
overlap)?  Rehearses the N-GPU path of bench.py (--sharding rows) without the gather.

  python tools/chunk_overlap.py [--workload C5] [--world 8] [--rank 0] [--chunks 4] [--reps 10]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ray-tracer-challenge-rust_amd"))

import numpy as np  # noqa: E402

import rtc_amd  # noqa: E402
from rtc_amd import abi, scenes  # noqa: E402
from rtc_amd.sharding import RowShard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C5")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--chunks", default="1,2,4")
    ap.add_argument("--streams", default="1,2,4")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cam, world, L = scenes.CONFIGS[args.workload](rtc_amd.api)
    sc = world.upload(0)
    desc = cam.desc()
    B = 16
    shard = RowShard(cam.vsize, B, args.rank, args.world)
    buf = torch.zeros((shard.padded_rows, cam.hsize, 3), dtype=torch.float32, device=dev)
    full = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    lo_pri, hi_pri = torch.cuda.Stream.priority_range()  # (least, greatest) = (0, -1) or wider
    pri_streams = [torch.cuda.Stream(device=dev, priority=hi_pri), torch.cuda.Stream(device=dev, priority=lo_pri)]

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), float(min(ts))

    def whole():
        sc.render_device(desc, abi.RtcRenderOpts(L, 0, 0, 0, 0), full.data_ptr(), 0, streams[0].cuda_stream)

    m, lo = timed(whole)
    print(f"{args.workload}: whole frame on one device {m:.4f} ms (min {lo:.4f}); 1/{args.world} of it = {m / args.world:.4f} ms")
    for chunks in [int(c) for c in args.chunks.split(",")]:
        cargs = [shard.chunk_render_args(c, chunks) for c in range(chunks)]
        for ns in [int(s) for s in args.streams.split(",")]:
            if ns > chunks:
                continue

            def run():
                for c, (first, count, row0) in enumerate(cargs):
                    if count:
                        o = abi.RtcRenderOpts(L, B, first, args.world, 0, count)
                        sc.render_device(desc, o, buf[row0:].data_ptr(), 0, streams[c % ns].cuda_stream)

            m, lo = timed(run)
            print(f"  rank {args.rank}/{args.world}: {chunks} chunk(s) on {ns} stream(s): {m:.4f} ms (min {lo:.4f})", flush=True)
    # chunks alternating between a high- and a low-priority stream: the dispatcher prefers the earlier chunk's workgroups, so it
    # drains while the next one already fills the machine, and finishes early enough for its gather to overlap
    for chunks in [int(c) for c in args.chunks.split(",")]:
        if chunks < 2:
            continue
        cargs = [shard.chunk_render_args(c, chunks) for c in range(chunks)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]

        def run_pri(record=False):
            if record:
                evs[0].record(pri_streams[0])
                pri_streams[1].wait_event(evs[0])
            for c, (first, count, row0) in enumerate(cargs):
                st = pri_streams[c % 2]
                if count:
                    o = abi.RtcRenderOpts(L, B, first, args.world, 0, count)
                    sc.render_device(desc, o, buf[row0:].data_ptr(), 0, st.cuda_stream)
                if record:
                    evs[c + 1].record(st)

        m, lo = timed(run_pri)
        torch.cuda.synchronize()
        run_pri(record=True)
        torch.cuda.synchronize()
        done = [evs[0].elapsed_time(evs[c + 1]) for c in range(chunks)]
        print(f"  rank {args.rank}/{args.world}: {chunks} chunks alternating high / low priority streams (priorities {hi_pri}, {lo_pri}): "
              f"{m:.4f} ms (min {lo:.4f}); chunk completion times " + ", ".join(f"{d:.3f}" for d in done), flush=True)
    sc.close()


if __name__ == "__main__":
    main()
