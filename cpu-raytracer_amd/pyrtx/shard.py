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
            # one contiguous receive buffer (rank r's tiles at [r*n, (r+1)*n)) and ONE scatter into raster order per frame
            n = len(mine)
            self.recv_buf = torch.zeros(world * n, dtype=torch.int32, device=device)
            self.recv = [self.recv_buf[r * n:(r + 1) * n] for r in range(world)]
            allidx = [tile_pixel_indices(width, height, r, world) for r in range(world)]
            self.valid = torch.from_numpy(np.concatenate([np.nonzero(a >= 0)[0] + r * n for r, a in enumerate(allidx)])).to(device)
            self.dest = torch.from_numpy(np.concatenate([a[a >= 0] for a in allidx])).to(device)
            self.all_valid = int(self.valid.numel()) == world * n
            self.final = torch.zeros(width * height, dtype=torch.int32, device=device)

    def __call__(self, packed_flat):
        """packed_flat: this rank's width*height int32 framebuffer (only its own tiles rendered)."""
        import torch
        import torch.distributed as dist
        torch.index_select(packed_flat, 0, self.my_idx, out=self.send)
        dist.gather(self.send, self.recv if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            src = self.recv_buf if self.all_valid else self.recv_buf.index_select(0, self.valid)
            self.final.index_copy_(0, self.dest, src)
            return self.final
        return None
