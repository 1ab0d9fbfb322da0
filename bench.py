#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X renderer (contract: see the task's bench.py section).

A "step" is one Camera::render of one frame with the flattened scene already resident in HBM.
metric = Mrays/s, ray = one World::intersect call (radiance + shadow rays).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C4|C5|C2] [--sharding rows|frames]
                  [--exact-sweep] [--no-cpu-baseline] [--no-parity] [--no-extra]

N = 1: BASELINE.json's quoted configuration, C3 (`objs/teapot.obj`, 6 320 triangles, the reference's
  putting_it_together_teapot scene at 1920x1080, RECURSION_LIMIT 5).
N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): BASELINE.json's multi-GPU configuration C5
  (`objs/pumpkin_tall_10k.obj`, glass, 3840x2160, depth 8; --workload C4 selects the cow), ONE frame per step: the frame's rows are dealt to the ranks in
  interleaved 16-row blocks (RtcRenderOpts.block_*), every rank renders its blocks in two chunks on a high- and a low-priority
  stream and the first chunk is gathered to rank 0 over xGMI while the second renders; rank 0 de-interleaves into the Canvas.  All of that is
  inside the timed region: "scaling": "strong", value = rays of the one frame / time.
  --sharding frames: every rank renders whole frames, no data-path collective (replicas; reported as `frames_mode`
  next to the headline when not selected, never as `value` by default).

The JSON line carries
  roofline      the render kernel against the vector-ALU peak: a divergent traversal, so the binding roof is
                VALU issue, not HBM and not MFMA; the compulsory-HBM fraction, the PMC-measured HBM
                traffic and the counter-derived VALU-busy / occupancy figures (newest profiles/rNN) are inside it.
  cpu_baseline  the CPU oracle in faithful-cost mode (per-ray cofactor inverse + per-ray Bounds::new: the work
                the single-threaded reference does), timed on a strided pixel subsample of the same frame.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch  # first: librtc_hip.so must share torch's HIP runtime (see rtc_amd/__init__.py)
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "ray-tracer-challenge-rust_amd"), os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector), FMA = 2 flop; vector FP64 runs at half of it
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOP_PER_TRIANGLE_TEST = 45      # full Möller–Trumbore evaluation as written in shape.rs:438-459
FLOP_PER_PRIMITIVE_TEST = 30     # sphere ~30 flop (plane ~2)
FLOP_PER_BOX_TEST = 21           # 6 sub + 6 mul + 6 min/max + 3 compares (rtc_device.hpp::box_filter)
# what the frame is rendered from: no RNG anywhere; the meshes are the reference's own model data (vertex / face
# arrays of /root/reference/objs/*.obj stored as assets/meshes/*.npz), the scenes are main.rs's or SURVEY.md §8d's
DATA_NOTE = {
    "C2": "synthetic: build-authored book-cover scene (SURVEY.md 8d), deterministic, no RNG",
    "C3": "reference model data: objs/teapot.obj as assets/meshes/teapot.npz in the reference's teapot scene (main.rs:368-397); deterministic, no RNG",
    "C4": "reference model data: objs/cow-nonormals.obj as assets/meshes/cow.npz in the reference's cow scene (main.rs:328-363); deterministic, no RNG",
    "C5": "reference model data: objs/pumpkin_tall_10k.obj as assets/meshes/pumpkin.npz in a build-authored glass scene (SURVEY.md 8d); deterministic, no RNG",
}
DATA_NOTE["C3zoom"] = DATA_NOTE["C3"]
C3ZOOM_VIEW = ((0.0, 1.2, -4.2), (0.2, 0.1, 0.0), (0.0, 1.0, 0.0))  # the C3 scene with the teapot filling the frame


def newest_profile(name):
    """profiles/rNN/<name> of the latest round that has it (committed rocprofv3 summaries), or None."""
    root = os.path.join(REPO, "profiles")
    for d in sorted((x for x in os.listdir(root) if x.startswith("r") and x[1:].isdigit()), reverse=True) if os.path.isdir(root) else []:
        f = os.path.join(root, d, name)
        if os.path.exists(f):
            return f
    return None


GATHER_NAME = {"nccl": "RCCL gather over xGMI", "gloo": "gloo gather through the host (rehearsal)"}
BLOCK_ROWS = 16                  # row-block height for row sharding = the kernel's pixel-tile height


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=["C2", "C3", "C4", "C5", "C3zoom"],
                    help="default: C3 on one GPU, C5 (BASELINE.json's configs[4], an 8-GPU configuration) on several")
    ap.add_argument("--width", type=int, default=0, help="override hsize (default: the config's)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--sharding", default="rows", choices=["rows", "frames"])
    ap.add_argument("--chunks", type=int, default=2, help="rows mode: pieces a rank's blocks are rendered and gathered in "
                    "(on streams of alternating priority; tools/chunk_overlap.py measures the choice)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary cases (C3zoom; frames mode for N > 1)")
    ap.add_argument("--exact-sweep", action="store_true",
                    help="brute force (RTC_FLAG_EXACT_SWEEP): every ray x every leaf in f64, LDS-staged triangles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on a one-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE={world})")
            return 2
        args.gpus = world
    if args.workload is None:
        args.workload = "C3" if world == 1 else "C5"
    if not torch.cuda.is_available():
        log("bench.py: no GPU visible; the renderer has no CPU fallback")
        return 3
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    import __graft_entry__ as ge

    if rank == 0:
        ge.ensure_built()
    if world > 1:
        dist.barrier()

    import rtc_amd
    from rtc_amd import abi, scenes
    from rtc_amd.sharding import CanvasGather, RowShard

    lib = rtc_amd.hip()
    size_kw = {}
    if args.width:
        size_kw["hsize"] = args.width
    if args.height:
        size_kw["vsize"] = args.height
    def make_scene(api):
        if args.workload == "C3zoom":  # secondary case (extra_cases.C3zoom; profiled on its own by tools/profile_round.sh C3zoom)
            c, ww, ll = scenes.c3_teapot(api, **size_kw)
            c.set_transform(api.view_transform(*C3ZOOM_VIEW))
            return c, ww, ll
        return scenes.CONFIGS[args.workload](api, **size_kw)

    cam, w, L = make_scene(rtc_amd.api)
    H, W = cam.vsize, cam.hsize
    info = w.flat_info()
    scene = w.upload(local_rank)  # flatten + H2D once: inputs are resident in HBM before the timed region
    desc = cam.desc()
    stream = torch.cuda.current_stream()
    xflag = abi.RTC_FLAG_EXACT_SWEEP if args.exact_sweep else 0
    rows_mode = world > 1 and args.sharding == "rows"

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if world > 1:
            tt = t.to(cdev)
            dist.all_reduce(tt, op=op)
            t = tt.cpu()
        return t

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the two ways a frame is produced ------------------------------------------------------------------
    full_opts = abi.RtcRenderOpts(L, 0, 0, 0, xflag)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)  # this rank's whole frame (frames mode, N = 1)
    cg = None
    if world > 1:
        shard = RowShard(H, BLOCK_ROWS, rank, world)
        assert int(lib.rtc_local_rows(H, C.byref(abi.RtcRenderOpts(L, BLOCK_ROWS, rank, world, xflag)))) == shard.local_rows
        cg = CanvasGather(shard, W, cdev, chunks=args.chunks)
        local_rows_buf = cg.local if cdev == dev else torch.zeros_like(cg.local, device=dev)
        chunk_args = [shard.chunk_render_args(c, cg.chunks) for c in range(cg.chunks)]
        # Chunks alternate between a high- and a low-priority stream: the dispatcher prefers the earlier chunk's workgroups, so
        # its long-lived waves drain while the next chunk already fills the machine, and its gather overlaps the rest of the render.
        # (Chunks queued on one stream each wait for the previous chunk's last wave: an eighth of C5's rows takes 0.75 ms in one
        # launch, 1.92 ms as four launches on one stream, 0.75 ms as two launches on a high / low pair with the first complete
        # after 0.52 ms; profiles/r02/v23_*.)
        pri_least, pri_greatest = torch.cuda.Stream.priority_range()
        rstreams = ([torch.cuda.Stream(device=dev, priority=pri_greatest), torch.cuda.Stream(device=dev, priority=pri_least)]
                    if cg.chunks > 1 else [stream])

    def step_frames(flags=0):
        o = abi.RtcRenderOpts(L, 0, 0, 0, xflag | flags)
        scene.render_device(desc, o, frame.data_ptr(), 0, stream.cuda_stream)

    def step_rows(flags=0, ev=None):
        """ONE frame: this rank's interleaved row blocks, rendered chunk by chunk; every finished chunk is handed to
        an asynchronous gather to rank 0 (RCCL runs it on its own stream, behind the render stream's work so far),
        so it travels over xGMI while the next chunk renders; rank 0 then de-interleaves into the Canvas."""
        if ev is not None:
            ev[0].record(stream)
        for rs in rstreams:
            if rs is not stream:
                rs.wait_stream(stream)  # behind the previous frame's de-interleave, which read the buffers these launches write
        for c, (first, count, row0) in enumerate(chunk_args):
            rs = rstreams[c % len(rstreams)]
            if count:
                o = abi.RtcRenderOpts(L, BLOCK_ROWS, first, world, xflag | flags, count)
                scene.render_device(desc, o, local_rows_buf[row0:].data_ptr(), 0, rs.cuda_stream)
            with torch.cuda.stream(rs):  # the gather of chunk c waits for this stream's work only
                if cg.local is not local_rows_buf:  # gloo rehearsal on one GPU: stage through the host
                    j0, j1 = shard.chunk(c, cg.chunks)
                    cg.local[j0 * BLOCK_ROWS: j1 * BLOCK_ROWS].copy_(local_rows_buf[j0 * BLOCK_ROWS: j1 * BLOCK_ROWS])
                    rs.synchronize()
                cg.send_chunk(c)
        for rs in rstreams:
            if rs is not stream:
                stream.wait_stream(rs)
        if ev is not None:
            ev[1].record(stream)  # all of this rank's render launches are complete here
        cg.finish()

    step = step_rows if rows_mode else step_frames

    # ---- counted pass (outside the timed region): rays and leaf tests per step, deterministic ---------------
    scene.reset_counters()
    step(abi.RTC_FLAG_COUNT_RAYS)
    sync()
    st = scene.stats()
    counts = torch.tensor([st["radiance_rays"], st["shadow_rays"], st["triangle_tests"], st["primitive_tests"],
                           st["filter_tests"], st["box_tests"], st["culled_rays"], st["unswept_shadow_rays"]], dtype=torch.float64)

    def flops(c):  # (f64 flop, f32 flop) of the leaf tests a launch executes
        return (c[2] * FLOP_PER_TRIANGLE_TEST + c[3] * FLOP_PER_PRIMITIVE_TEST,
                c[4] * FLOP_PER_TRIANGLE_TEST + c[5] * FLOP_PER_BOX_TEST)

    lf64, lf32 = flops(counts.tolist())
    local_flop = lf32 + 2.0 * lf64  # FP32-equivalent: one f64 flop occupies the vector ALU like two f32 flops
    counts = all_reduce(counts)
    radiance, shadow, tri_tests, prim_tests, filter_tests, box_tests, culled, unswept = [int(x) for x in counts.tolist()]
    rays_per_step = radiance + shadow  # whole job: all ranks

    # ---- warmup + timed region -----------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    sync()
    t0 = time.perf_counter()
    if rows_mode:
        for k in range(args.steps):
            step_rows(ev=ev[k])  # HIP events around this rank's render launches, on their stream
    else:
        # ONE pair of HIP events over the timed region, on the stream the kernel is launched on: the K dependent launches follow one
        # another without anything between them.  (Two events per launch, as in rounds 1-3, put 5 us of event packets between the
        # frames -- 0.0786 against 0.0734 ms per C3 frame, tools/launch_modes.py -- and that was 7 % of `value`.)
        region[0].record(stream)
        for k in range(args.steps):
            scene.render_device(desc, full_opts, frame.data_ptr(), 0, stream.cuda_stream)
        region[1].record(stream)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = float(all_reduce(torch.tensor([elapsed], dtype=torch.float64), dist.ReduceOp.MAX).item())
    kernel_ms_each = None
    if rows_mode:
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else 0.0
    else:
        kernel_ms = region[0].elapsed_time(region[1]) / args.steps if args.steps else 0.0
        # outside the timed region: the same launches with an event pair around each (what rocprofv3's per-kernel duration corresponds to)
        for k in range(args.steps):
            ev[k][0].record(stream)
            scene.render_device(desc, full_opts, frame.data_ptr(), 0, stream.cuda_stream)
            ev[k][1].record(stream)
        sync()
        kernel_ms_each = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else 0.0
    kflop = float(all_reduce(torch.tensor([local_flop / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0], dtype=torch.float64)).item())
    kms = float(all_reduce(torch.tensor([kernel_ms], dtype=torch.float64), dist.ReduceOp.MAX).item())

    # ---- N > 1: the other sharding as a secondary figure (never `value` unless selected) -----------------------
    other = None
    if world > 1 and not args.no_extra:
        o_steps = max(5, min(20, args.steps))
        o_step = step_frames if rows_mode else step_rows
        for _ in range(2):
            o_step()
        sync()
        t1 = time.perf_counter()
        for _ in range(o_steps):
            o_step()
        sync()
        o_elapsed = float(all_reduce(torch.tensor([time.perf_counter() - t1], dtype=torch.float64), dist.ReduceOp.MAX).item())
        one_frame = rays_per_step if rows_mode else rays_per_step // world
        if rows_mode:
            other = ("frames_mode", {"ms_per_step": round(o_elapsed / o_steps * 1e3, 4), "steps": o_steps,
                                     "Mrays_s": round(one_frame * world * o_steps / o_elapsed / 1e6, 1),
                                     "what": f"{world} whole frames per step, one per rank, no data-path collective: replicated "
                                             f"work (weak scaling), not the north-star's sharding"})
        else:
            other = ("row_sharded", {"ms_per_frame": round(o_elapsed / o_steps * 1e3, 4), "steps": o_steps,
                                     "Mrays_s": round(one_frame * o_steps / o_elapsed / 1e6, 1),
                                     "what": f"ONE frame per step: interleaved {BLOCK_ROWS}-row blocks over {world} ranks, "
                                             f"gathered to rank 0 ({GATHER_NAME[args.backend]}), strong scaling"})
        if rank == 0:
            other[1]["identical_frames"] = bool(torch.equal(cg.canvas.to(dev), frame))
    # the same frame on ONE GPU (rank 0 alone, the others wait): what the row-sharded figure is a speed-up over
    one_gpu = None
    if world > 1 and not args.no_extra:
        sync()
        if rank == 0:
            for _ in range(2):
                step_frames()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            n1 = max(3, min(10, args.steps))
            for _ in range(n1):
                step_frames()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t2) / n1
            one = rays_per_step if rows_mode else rays_per_step // world
            one_gpu = {"ms_per_frame": round(dt * 1e3, 4), "Mrays_s": round(one / dt / 1e6, 1),
                       "what": "the same frame rendered by rank 0 alone, scene resident, no gather"}
        sync()

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    canvas = cg.canvas if rows_mode else frame
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_per_step * args.steps / elapsed / 1e6
    achieved_tflops = kflop / 1e12
    peak = FP32_VECTOR_PEAK_TFLOPS * world
    frames_per_step = 1 if (world == 1 or rows_mode) else world
    f64_flop, f32_flop = flops([0, 0, tri_tests, prim_tests, filter_tests, box_tests])
    compulsory_bytes = frames_per_step * ((info["n_triangles"] * 96 + info["n_leaves"] * 64) + H * W * 12)
    hbm_gbs = compulsory_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    doc = ("the C3 scene with the camera moved in until the teapot fills the frame" if args.workload == "C3zoom"
           else scenes.CONFIGS[args.workload].__doc__.strip().splitlines()[0])
    if world == 1:
        sharding = "single GPU"
    elif rows_mode:
        sharding = (f"one frame per step: interleaved {BLOCK_ROWS}-row blocks over {world} ranks, rendered in {cg.chunks} chunks "
                    f"on streams of alternating priority, each gathered to rank 0 ({GATHER_NAME[args.backend]}) while the next renders")
    else:
        sharding = f"{world} frames per step, one per rank, no data-path collective"
    swept = rays_per_step - culled - unswept
    out = {
        "metric": f"Mrays/s (primary+secondary) @{W}x{H} depth{L}; max|dcolor| vs CPU",
        "value": round(mrays, 3),
        # the same clock, counting only the rays that cost a sweep (no wave-culled primaries, no unswept shadow rays): the
        # figure to compare kernels by; `value` counts every World::intersect call, as the metric defines a ray
        "value_swept": round(swept * args.steps / elapsed / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if (rows_mode or world == 1) else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": DATA_NOTE[args.workload],
        "config": {"workload": f"{args.workload}: {doc}", "resolution": [W, H], "recursion_limit": L,
                   "triangles": info["n_triangles"], "leaves": info["n_leaves"], "sharding": sharding},
        # the headline counts every World::intersect call; the split says how many of them cost a sweep
        "rays_per_step": {"radiance": radiance, "shadow": shadow, "swept": swept, "culled_primary": culled,
                          "unswept_shadow": unswept,
                          "note": "culled_primary = primary rays of waves whose 64 rays all miss every top-level group box "
                                  "(answered BLACK before any sweep); unswept_shadow = shadow rays of hits that face away from the "
                                  "light, whose answer Material::lighting ignores (material.rs:57-60): cast and counted as in the "
                                  "reference, not swept; swept = all the others"},
        "Mrays_s_split": {"swept": round(swept * args.steps / elapsed / 1e6, 3), "culled_primary": round(culled * args.steps / elapsed / 1e6, 3),
                          "unswept_shadow": round(unswept * args.steps / elapsed / 1e6, 3)},
        "sweep": "exact brute force (LDS-staged)" if args.exact_sweep else "accelerated (clusters + f32 filters + exact f64 confirm)",
        "leaf_tests_per_step": {"triangle_exact_f64": tri_tests, "primitive_f64": prim_tests,
                                "triangle_filter_f32": filter_tests, "box_f32": box_tests},
        "roofline": {
            "bound": "valu",
            "achieved": round(achieved_tflops, 6),
            "peak": round(peak, 1),
            "unit": "TFLOP/s",
            "frac": round(round(achieved_tflops, 6) / round(peak, 1), 8),  # the quotient of the two printed figures
            "traffic": None,
            "kernel_ms": round(kms, 4),
            "kernel_ms_what": ("mean over the steps of an event pair around each rank's render launches" if rows_mode else
                               "one HIP event pair over the K dependent launches of the timed region, on their stream, / K"),
            "kernel_ms_event_pair_per_launch": round(kernel_ms_each, 4) if kernel_ms_each is not None else None,
            "flop_per_step": {"f64": int(f64_flop), "f32": int(f32_flop)},
            "reference_algorithm_tflops": round(swept / frames_per_step * info["n_triangles"] * FLOP_PER_TRIANGLE_TEST
                                                / (kms * 1e-3) / 1e12, 3) if kms > 0 else None,
            "note": "achieved = FP32-equivalent algorithmic flop of the leaf tests the launch executes (45 per triangle "
                    "test, 30 per primitive, 21 per box; an f64 flop counts as two) / kernel time (HIP events on the launch "
                    "stream), against the vector FP32 peak (FP64 peak is half of it).  The exact f64 tests cannot use FMA "
                    "(parity), which alone caps them at half of their roof.  reference_algorithm_tflops = swept rays x all "
                    "triangles x 45 / time: what a brute-force sweep would have to sustain for the same swept rays/s.  traffic and "
                    "the `pmc` / `valu_issue` figures come from rocprofv3 PMC passes of this very command "
                    "(tools/profile_pmc.sh, summaries under profiles/), not from inside bench.py.",
            "hbm": {"compulsory_bytes": int(compulsory_bytes), "achieved": round(hbm_gbs, 3), "peak": HBM_PEAK_GBS * world,
                    "unit": "GB/s", "frac": round(hbm_gbs / (HBM_PEAK_GBS * world), 6)},
        },
    }
    if other:
        out[other[0]] = other[1]
    if one_gpu:
        out["one_gpu_same_workload"] = one_gpu
        if rows_mode:
            out["speedup_over_one_gpu"] = round(one_gpu["ms_per_frame"] / ms_per_step, 3)
    # HBM traffic per launch and the utilisation figures come from PMC counters, which need rocprofv3
    # (tools/profile_pmc.sh); the committed summary of that run on this very command is quoted here (FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is, separate passes).
    pmc = newest_profile(f"pmc_{args.workload}_summary.json")
    if world == 1 and not size_kw and not args.exact_sweep and pmc:
        try:
            with open(pmc) as fh:
                ps = json.load(fh)
            out["roofline"]["traffic"] = int(ps["hbm_traffic_bytes_per_launch"])
            out["roofline"]["traffic_source"] = f"{os.path.relpath(pmc, REPO)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
            dv = ps.get("derived", {})
            keys = ("valu_busy", "mean_waves_per_simd", "valu_lane_utilisation", "valu_active_share_of_wave_cycles",
                    "wait_any_share", "l2_hit_rate", "effective_clock_ghz")
            out["roofline"]["pmc"] = {k: round(dv[k], 4) for k in keys if k in dv}
            out["roofline"]["pmc"]["note"] = ("valu_busy = SQ_ACTIVE_INST_VALU*4 / 1024 SIMDs / (GRBM_GUI_ACTIVE/8); "
                                              "mean_waves_per_simd = SQ_WAVE_CYCLES*4 / 1024 / (GRBM_GUI_ACTIVE/8); all from one "
                                              "PMC pass of the profiled run (kernel time there: see kernel_stats in the summary)")
            if "valu_issue" in ps:
                out["roofline"]["valu_issue"] = ps["valu_issue"]
        except Exception as e:  # a malformed summary must not cost the bench line
            out["roofline"]["traffic_source"] = f"unreadable {pmc}: {e}"

    # ---- a second case for C3: the mesh fills the frame (no background to average over) -------------------------
    if world == 1 and args.workload == "C3" and not size_kw and not args.exact_sweep and not args.no_extra:
        zc, zw, zL = scenes.c3_teapot(rtc_amd.api)
        zc.set_transform(rtc_amd.api.view_transform(*C3ZOOM_VIEW))
        zs = zw.upload(local_rank)
        zd = zc.desc()
        zs.reset_counters()
        zs.render_device(zd, abi.RtcRenderOpts(zL, 0, 0, 0, abi.RTC_FLAG_COUNT_RAYS), frame.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        zst = zs.stats()
        zo = abi.RtcRenderOpts(zL, 0, 0, 0, 0)
        for _ in range(3):
            zs.render_device(zd, zo, frame.data_ptr(), 0, stream.cuda_stream)
        za, zb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        za.record(stream)
        for _ in range(20):
            zs.render_device(zd, zo, frame.data_ptr(), 0, stream.cuda_stream)
        zb.record(stream)
        torch.cuda.synchronize()
        zms = za.elapsed_time(zb) / 20
        zrays = zst["radiance_rays"] + zst["shadow_rays"]
        out["extra_cases"] = {"C3zoom": {
            "what": "the C3 scene with the camera moved in until the teapot fills the 1920x1080 frame (view_transform from "
                    "(0,1.2,-4.2) to (0.2,0.1,0)): almost no wave-culled background",
            "ms_per_frame": round(zms, 4), "rays": zrays, "culled_primary": zst["culled_rays"],
            "Mrays_s": round(zrays / zms / 1e3, 1)}}
        zpmc = newest_profile("pmc_C3zoom_summary.json")  # counters of this case, when a committed profile of it exists
        if zpmc:
            try:
                with open(zpmc) as fh:
                    zdv = json.load(fh).get("derived", {})
                out["extra_cases"]["C3zoom"]["pmc"] = {k: round(zdv[k], 4) for k in ("valu_busy", "mean_waves_per_simd", "valu_lane_utilisation") if k in zdv}
                out["extra_cases"]["C3zoom"]["pmc"]["source"] = os.path.relpath(zpmc, REPO) + " (rocprofv3 PMC passes of `bench.py --workload C3zoom`)"
            except Exception:
                pass
        zs.close()
        # Secondary figure, never `value`: the same C3 frames double-buffered on two streams, so that one frame's ramp fills the
        # other's tail (what a host that renders a sequence of frames can do; the timed region above launches dependent frames
        # on one stream).
        two = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        fb2 = [frame, torch.zeros_like(frame)]
        for s2 in two:
            s2.wait_stream(stream)
        for k in range(4):
            scene.render_device(desc, full_opts, fb2[k % 2].data_ptr(), 0, two[k % 2].cuda_stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for k in range(40):
            scene.render_device(desc, full_opts, fb2[k % 2].data_ptr(), 0, two[k % 2].cuda_stream)
        torch.cuda.synchronize()
        pms = (time.perf_counter() - t2) * 1e3 / 40
        out["extra_cases"]["two_frames_in_flight"] = {
            "what": f"{args.workload} frames alternating between two streams and two framebuffers (40 frames, wall clock)",
            "ms_per_frame": round(pms, 4), "Mrays_s": round(rays_per_step / pms / 1e3, 1)}
        # re-render the timed frame for the parity check below (the buffer was reused)
        scene.render_device(desc, full_opts, frame.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()

    # ---- parity spot check of the very frame that was timed (outside the timed region) -----------------------
    img = canvas.cpu().numpy()
    if not args.no_parity:
        import oracle_binding

        co, wo, _ = make_scene(oracle_binding.api)
        xs, ys = np.meshgrid(np.arange(8, W, 16), np.arange(8, H, 16))
        xs, ys = xs.ravel(), ys.ravel()
        ref, _oinfo = co.render_pixels(wo, xs, ys, L, mode=oracle_binding.MODE_CACHED, nthreads=min(16, os.cpu_count() or 1))
        d = float(np.abs(img[ys, xs].astype(np.float64) - ref).max())
        out["max_abs_dcolor"] = d
        out["parity"] = {"pixels_checked": int(xs.size), "tolerance": 1e-5, "ok": bool(d < 1e-5),
                         "what": "f32 framebuffer vs CPU oracle (f64) on every 16th pixel in x and y, offset 8"}

    # ---- CPU baseline: the oracle in the reference's cost model, one core, bounded sample ---------------------
    if world == 1 and not args.no_cpu_baseline:
        import oracle_binding

        co, wo, _ = make_scene(oracle_binding.api)
        # calibrate the stride on a tiny sample, then run ~cpu-seconds of single-threaded work
        xs, ys = np.meshgrid(np.arange(96, W, 192), np.arange(96, H, 192))
        _, cal = co.render_pixels(wo, xs.ravel(), ys.ravel(), L, mode=oracle_binding.MODE_FAITHFUL, nthreads=1)
        per_px = cal["seconds"] / max(1, xs.size)
        npx = max(16, int(args.cpu_seconds / max(per_px, 1e-9)))
        stride = max(1, int(np.sqrt(W * H / npx)))
        xs, ys = np.meshgrid(np.arange(stride // 2, W, stride), np.arange(stride // 2, H, stride))
        xs, ys = xs.ravel(), ys.ravel()
        _, fi = co.render_pixels(wo, xs, ys, L, mode=oracle_binding.MODE_FAITHFUL, nthreads=1)
        _, ci = co.render_pixels(wo, xs, ys, L, mode=oracle_binding.MODE_CACHED, nthreads=1)
        f_rays = fi["radiance_rays"] + fi["shadow_rays"]
        out["cpu_baseline"] = {
            "value": round(f_rays / fi["seconds"] / 1e6, 9),
            "unit": "Mrays/s",
            "cores": 1,
            "kind": "port",
            "sample": f"every {stride}th pixel in x and y (offset {stride // 2}) of the same {W}x{H} frame: {xs.size} pixels, "
                      f"{f_rays} rays, {fi['seconds']:.1f} s; faithful-cost mode (per-ray cofactor inverse per leaf + per-ray "
                      f"Bounds::new, as shape.rs:249-253,401)",
            "cached_inverse_value": round((ci["radiance_rays"] + ci["shadow_rays"]) / ci["seconds"] / 1e6, 6),
            "host_cpus": os.cpu_count(),
        }
        # the CPU sweeps every ray it casts: the comparable GPU figure is the rate of swept rays (value_swept), not `value`
        out["speedup_vs_cpu_reference_cost"] = round(out["value_swept"] / out["cpu_baseline"]["value"], 1)
        out["speedup_vs_cpu_reference_cost_all_rays"] = round(mrays / out["cpu_baseline"]["value"], 1)

    print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
