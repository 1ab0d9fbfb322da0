"""Row sharding of one frame across the GPUs of a node (one process per GPU, torch.distributed over RCCL).

Pixels never interact (camera.rs:70-76) and the flattened scene is small, so the scene is replicated and the
image rows are dealt to the ranks in interleaved blocks (row cost is very uneven: background rows are one cheap
ray per pixel, mesh rows 2-14).  The only exchange step of the path is the gather of the Canvas slices to rank 0;
there is no other collective.  The C ABI renders "blocks block_first, block_first+block_step, ... (block_count of
them)" densely into a local buffer (RtcRenderOpts, include/rtc_hip.h); this module holds the matching host-side
bookkeeping.

A rank's blocks are rendered in a few chunks and every finished chunk is handed to an asynchronous gather at once:
RCCL runs it on its own stream behind an event of the render stream, so chunk k travels over xGMI while chunk k+1
renders (rtc_multi_render does the same inside one process).  Rank 0 puts the blocks back in Canvas order with one
strided copy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class RowShard:
    vsize: int
    block_rows: int
    rank: int
    world: int

    @property
    def n_blocks(self) -> int:
        return (self.vsize + self.block_rows - 1) // self.block_rows

    def blocks(self, rank: Optional[int] = None) -> range:
        return range(self.rank if rank is None else rank, self.n_blocks, self.world)

    def rows(self, rank: Optional[int] = None) -> List[int]:
        """Global row index of every local row of `rank`, in local (dense) order."""
        out: List[int] = []
        for bi in self.blocks(rank):
            out.extend(range(bi * self.block_rows, min(self.vsize, (bi + 1) * self.block_rows)))
        return out

    @property
    def local_rows(self) -> int:
        """Same number rtc_local_rows() returns for these options."""
        return len(self.rows())

    @property
    def padded_blocks(self) -> int:
        """Blocks of the largest shard (every rank's gather buffer holds this many)."""
        return (self.n_blocks + self.world - 1) // self.world

    @property
    def padded_rows(self) -> int:
        return self.padded_blocks * self.block_rows

    def chunk(self, c: int, chunks: int) -> Tuple[int, int]:
        """Local blocks [j0, j1) of chunk c: the same split of 0..padded_blocks on every rank, so that the chunk
        messages have equal sizes; a rank with fewer blocks simply renders fewer of them."""
        nb = self.padded_blocks
        return nb * c // chunks, nb * (c + 1) // chunks

    def chunk_render_args(self, c: int, chunks: int) -> Tuple[int, int, int]:
        """(block_first, block_count, first local row) to render chunk c with rtc_render_device: RtcRenderOpts(
        block_rows, block_first, block_step = world, block_count); block_count may be 0 (nothing to render)."""
        j0, j1 = self.chunk(c, chunks)
        mine = len(self.blocks())
        return self.rank + j0 * self.world, max(0, min(j1, mine) - min(j0, mine)), j0 * self.block_rows


class CanvasGather:
    """Gathers the ranks' dense row buffers to rank 0, chunk by chunk, and de-interleaves them into the Canvas layout
    (x + y*width, canvas.rs:20-26).  Works with any backend ("nccl" = RCCL on the GPUs, "gloo" in CPU tests)."""

    def __init__(self, shard: RowShard, hsize: int, device, dtype=torch.float32, group=None, chunks: int = 4):
        self.shard, self.hsize, self.group = shard, hsize, group
        self.chunks = max(1, min(chunks, shard.padded_blocks))
        nb, B, w = shard.padded_blocks, shard.block_rows, shard.world
        self.local = torch.zeros((nb * B, hsize, 3), dtype=dtype, device=device)
        self.is_root = shard.rank == 0
        self._pending: list = []
        if self.is_root:
            self._recv = torch.zeros((w, nb * B, hsize, 3), dtype=dtype, device=device)
            self._padded = torch.zeros((nb, w, B, hsize, 3), dtype=dtype, device=device)  # Canvas order, padded
            self.canvas = self._padded.view(nb * w * B, hsize, 3)[: shard.vsize]
        else:
            self._recv = self._padded = self.canvas = None

    def send_chunk(self, c: int):
        """Starts the gather of chunk c (its rows of `local` must already be queued on the current stream)."""
        if self.shard.world == 1:
            return
        j0, j1 = self.shard.chunk(c, self.chunks)
        if j1 <= j0:
            return
        B = self.shard.block_rows
        part = self.local[j0 * B: j1 * B]
        into = [self._recv[r, j0 * B: j1 * B] for r in range(self.shard.world)] if self.is_root else None
        self._pending.append(dist.gather(part, into, dst=0, group=self.group, async_op=True))

    def finish(self):
        """Waits for the chunks in flight; rank 0 assembles `canvas` (one strided copy) and returns it."""
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if not self.is_root:
            return None
        nb, B, w = self.shard.padded_blocks, self.shard.block_rows, self.shard.world
        if w == 1:
            self._padded.view(nb * B, self.hsize, 3).copy_(self.local)
        else:  # global block r + j*w is rank r's local block j
            self._padded.copy_(self._recv.view(w, nb, B, self.hsize, 3).permute(1, 0, 2, 3, 4))
        return self.canvas

    def gather(self):
        """The whole local buffer at once (every chunk, then finish)."""
        for c in range(self.chunks):
            self.send_chunk(c)
        return self.finish()


def render_sharded(render_rows: Callable[[RowShard, torch.Tensor], None], shard: RowShard, hsize: int, device,
                   group=None) -> Optional[torch.Tensor]:
    """Render one frame across the group: `render_rows(shard, local)` fills local[:shard.local_rows] with this
    rank's rows; returns the full canvas on rank 0 (None elsewhere)."""
    g = CanvasGather(shard, hsize, device, group=group)
    render_rows(shard, g.local)
    return g.gather()
