"""Row sharding of one frame across the GPUs of a node (one process per GPU, torch.distributed over RCCL).

Pixels never interact (camera.rs:70-76) and the flattened scene is small, so the scene is replicated and the
image rows are dealt to the ranks in interleaved blocks (row cost is very uneven: background rows are one cheap
ray per pixel, mesh rows 2-14).  The only exchange step of the path is the gather of the Canvas slices to rank 0;
there is no other collective.  The C ABI renders "blocks block_first, block_first+block_step, ..." densely into a
local buffer (RtcRenderOpts, include/rtc_hip.h); this module holds the matching host-side bookkeeping.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

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
    def padded_rows(self) -> int:
        """Rows of the equal-sized gather buffer (the largest shard, whole blocks)."""
        return ((self.n_blocks + self.world - 1) // self.world) * self.block_rows


class CanvasGather:
    """Gathers the ranks' dense row buffers to rank 0 and de-interleaves them into the Canvas layout
    (x + y*width, canvas.rs:20-26).  Works with any backend ("nccl" = RCCL on the GPUs, "gloo" in CPU tests)."""

    def __init__(self, shard: RowShard, hsize: int, device, dtype=torch.float32, group=None):
        self.shard, self.hsize, self.group = shard, hsize, group
        self.local = torch.zeros((shard.padded_rows, hsize, 3), dtype=dtype, device=device)
        self.is_root = shard.rank == 0
        self.canvas = torch.zeros((shard.vsize, hsize, 3), dtype=dtype, device=device) if self.is_root else None
        self._recv = [torch.zeros_like(self.local) for _ in range(shard.world)] if self.is_root else None
        self._dst = ([torch.tensor(shard.rows(r), dtype=torch.long, device=device) for r in range(shard.world)]
                     if self.is_root else None)

    def gather(self):
        """One collective: every rank contributes `local`; rank 0 assembles `canvas`."""
        if self.shard.world == 1:
            self.canvas.copy_(self.local[: self.shard.vsize])
            return self.canvas
        dist.gather(self.local, self._recv, dst=0, group=self.group)
        if self.is_root:
            for r in range(self.shard.world):
                idx = self._dst[r]
                self.canvas.index_copy_(0, idx, self._recv[r][: idx.numel()])
        return self.canvas


def render_sharded(render_rows: Callable[[RowShard, torch.Tensor], None], shard: RowShard, hsize: int, device,
                   group=None) -> Optional[torch.Tensor]:
    """Render one frame across the group: `render_rows(shard, local)` fills local[:shard.local_rows] with this
    rank's rows; returns the full canvas on rank 0 (None elsewhere)."""
    g = CanvasGather(shard, hsize, device, group=group)
    render_rows(shard, g.local)
    return g.gather()
