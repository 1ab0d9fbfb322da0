"""N > 1 path on CPU: world_size-2 (and 3) `gloo` runs of the row-sharding + Canvas-gather plumbing that bench.py
uses over RCCL on the GPUs.  The renderer itself needs a GPU, so each rank fills its rows with a deterministic
function of the GLOBAL pixel coordinates; rank 0 must end up with exactly the single-process canvas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtc_amd.sharding import CanvasGather, RowShard, render_sharded


def _pixel_value(y, x, c):
    return (y * 1000.0 + x + 0.25 * c).astype(np.float32)


def _expected(vsize, hsize):
    y, x, c = np.meshgrid(np.arange(vsize), np.arange(hsize), np.arange(3), indexing="ij")
    return _pixel_value(y, x, c)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, vsize, hsize, block_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = RowShard(vsize, block_rows, rank, world)

    def render_rows(sh, local):
        rows = np.array(sh.rows(), dtype=np.int64)
        assert len(rows) == sh.local_rows <= sh.padded_rows
        y, x, c = np.meshgrid(rows, np.arange(hsize), np.arange(3), indexing="ij")
        local[: len(rows)] = torch.from_numpy(_pixel_value(y, x, c))

    canvas = render_sharded(render_rows, shard, hsize, torch.device("cpu"))
    # a second frame through a persistent CanvasGather (bench.py's steady state)
    g = CanvasGather(shard, hsize, torch.device("cpu"))
    render_rows(shard, g.local)
    canvas2 = g.gather()
    # bench.py's timed loop: the shard rendered chunk by chunk, every chunk sent while the next one "renders"
    g3 = CanvasGather(shard, hsize, torch.device("cpu"), chunks=3)
    B = shard.block_rows
    for c in range(g3.chunks):
        first, count, row0 = shard.chunk_render_args(c, g3.chunks)
        if count:
            blocks = range(first, first + count * world, world)
            rows = np.array([y for b in blocks for y in range(b * B, min(vsize, (b + 1) * B))], dtype=np.int64)
            y, x, ch = np.meshgrid(rows, np.arange(hsize), np.arange(3), indexing="ij")
            g3.local[row0: row0 + len(rows)] = torch.from_numpy(_pixel_value(y, x, ch))
        g3.send_chunk(c)
    canvas3 = g3.finish()
    if rank == 0:
        assert torch.equal(canvas, canvas2) and torch.equal(canvas, canvas3)
        np.save(out_path, canvas.numpy().copy())
    else:
        assert canvas is None and canvas2 is None and canvas3 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,vsize,hsize,block_rows", [(2, 50, 24, 8), (2, 1080 // 8, 16, 16), (3, 37, 5, 4)])
def test_row_sharding_gather_gloo(tmp_path, world, vsize, hsize, block_rows):
    out = str(tmp_path / "canvas.npy")
    mp.spawn(_worker, args=(world, _free_port(), vsize, hsize, block_rows, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (vsize, hsize, 3)
    assert (got == _expected(vsize, hsize)).all()


def test_row_shard_bookkeeping_matches_the_abi():
    """RowShard must agree with rtc_local_rows() (include/rtc_hip.h) for every rank; no compute call is made."""
    import ctypes as C

    import rtc_amd
    from rtc_amd import abi

    lib = rtc_amd.hip()
    for vsize, B, world in [(1080, 16, 8), (2160, 16, 8), (50, 8, 3), (7, 16, 2), (16, 16, 4), (1, 1, 1)]:
        seen = []
        for r in range(world):
            sh = RowShard(vsize, B, r, world)
            opts = abi.RtcRenderOpts(5, B, r, world, 0)
            assert lib.rtc_local_rows(vsize, C.byref(opts)) == sh.local_rows
            seen += sh.rows()
        assert sorted(seen) == list(range(vsize))  # every row rendered exactly once
        # the same shard rendered in chunks (RtcRenderOpts.block_count): the chunks tile the shard, in order
        for chunks in (1, 3, 4):
            for r in range(world):
                sh = RowShard(vsize, B, r, world)
                got, row = [], 0
                for c in range(chunks):
                    first, count, row0 = sh.chunk_render_args(c, chunks)
                    if count == 0:
                        continue
                    assert row0 == row or row0 >= row
                    opts = abi.RtcRenderOpts(5, B, first, world, 0, count)
                    n = lib.rtc_local_rows(vsize, C.byref(opts))
                    blocks = range(first, first + count * world, world)
                    rows = [y for b in blocks for y in range(b * B, min(vsize, (b + 1) * B))]
                    assert n == len(rows)
                    got += rows
                    row = row0 + n
                assert got == sh.rows()
    assert lib.rtc_local_rows(1080, C.byref(abi.RtcRenderOpts(5, 0, 0, 0, 0))) == 1080
