"""Screen-tile sharding across the GPUs of one node (BASELINE.json configs[3]).

Tiles are numbered as WorkerThread.cpp:57-61 does (task -> x = (task % tile_count_x)*32, y = (task /
tile_count_x)*32).  GPU g of N owns tiles {t : t mod N == g}; after rendering, the only exchange is a
gather of the packed pixels of the owned tiles to rank 0, which un-swizzles tile-major -> raster.
The functions below are plain torch + numpy so the same code runs over RCCL on GPUs (bench.py) and
over gloo on CPU tensors (tests/test_shard_gloo.py).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

TILE = 32


def owned_tiles(tile_count: int, rank: int, world: int) -> range:
    return range(rank, tile_count, world)


def shard_args(tile_count: int, rank: int, world: int):
    """(first_tile, tile_stride, count) for rtx_render_tiles."""
    n = len(owned_tiles(tile_count, rank, world))
    return rank, world, n


def tile_pixel_indices(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """Raster pixel index of every slot of the rank's tile-major send buffer (-1 = padding:
    clipped pixels of edge tiles and the unused tail when tiles do not divide evenly)."""
    tcx = (width + TILE - 1) // TILE
    tcy = (height + TILE - 1) // TILE
    T = tcx * tcy
    per_rank = (T + world - 1) // world
    idx = np.full((per_rank, TILE, TILE), -1, np.int64)
    for k, t in enumerate(owned_tiles(T, rank, world)):
        ty, tx = (t // tcx) * TILE, (t % tcx) * TILE
        ys, xs = np.mgrid[ty:ty + TILE, tx:tx + TILE]
        idx[k] = np.where((ys < height) & (xs < width), ys * width + xs, -1)
    return idx.reshape(-1)


class TileGather:
    """Gathers the owned tiles of every rank's packed framebuffer into a full frame on rank 0."""

    def __init__(self, width: int, height: int, rank: int, world: int, device):
        import torch
        self.rank, self.world = rank, world
        mine = tile_pixel_indices(width, height, rank, world)
        self.my_idx = torch.from_numpy(np.where(mine >= 0, mine, 0)).to(device)
        self.send = torch.zeros(len(mine), dtype=torch.int32, device=device)
        if rank == 0:
            self.recv = [torch.zeros_like(self.send) for _ in range(world)]
            allidx = [tile_pixel_indices(width, height, r, world) for r in range(world)]
            self.valid = [torch.from_numpy(np.nonzero(a >= 0)[0]).to(device) for a in allidx]
            self.dest = [torch.from_numpy(a[a >= 0]).to(device) for a in allidx]
            self.final = torch.zeros(width * height, dtype=torch.int32, device=device)

    def __call__(self, packed_flat):
        """packed_flat: this rank's width*height int32 framebuffer (only its own tiles rendered)."""
        import torch
        import torch.distributed as dist
        torch.index_select(packed_flat, 0, self.my_idx, out=self.send)
        dist.gather(self.send, self.recv if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            for r in range(self.world):
                self.final.index_copy_(0, self.dest[r], self.recv[r].index_select(0, self.valid[r]))
            return self.final
        return None
