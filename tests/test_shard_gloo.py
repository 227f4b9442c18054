"""N > 1 path on CPU: world-size 2 / 3 gloo runs of the C side's tile partition and tile-major slot order.

What is tested is the index maths of librtx_hip.so itself — rtx_group_layout / rtx_group_slot_pixels are host wrappers around the
very functions k_resolve's tile-major writes and k_unswizzle use (csrc/rtx_trace.h: group_own_tiles, group_tile_of, tile_slot_pixel).
Each rank renders the tiles the library says it owns (the oracle stands in for the GPU: test infrastructure), fills its send buffer in
the library's slot order, one gather to rank 0 as rtx_group_render does it with ncclGather, and rank 0 assembles the frame the way
k_unswizzle does: slot i of rank r's part goes to the pixel the library names.  The frame must equal the golden frame."""
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "cpu-raytracer_amd"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import util, orc
    from pyrtx import api
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, g = util.load_golden(name)
    total, per_rank, own = api.group_layout(sc.width, sc.height, world, rank)
    assert total == sc.tile_count
    out = orc.OracleScene(sc).render(first_tile=rank, tile_stride=world, tile_count=own, threads=2)
    packed = out["packed"].reshape(-1).view(np.int32)
    slots = api.group_slot_pixels(sc.width, sc.height, world, rank)
    send = torch.from_numpy(np.where(slots >= 0, packed[np.maximum(slots, 0)], 0).astype(np.int32))
    recv = [torch.zeros(per_rank * 1024, dtype=torch.int32) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    stats = torch.tensor([out["stats"][k] for k in ("primary", "shadow", "reflection", "refraction")], dtype=torch.int64)
    dist.all_reduce(stats)
    if rank == 0:
        frame = np.zeros(sc.width * sc.height, np.int32)
        for r in range(world):
            sl = api.group_slot_pixels(sc.width, sc.height, world, r)
            frame[sl[sl >= 0]] = recv[r].numpy()[sl >= 0]
        ok = np.array_equal(frame.view(np.uint32).reshape(sc.height, sc.width), g["packed"])
        q.put((ok, stats.tolist() == g["stats"].tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("materials_aniso", 2), ("monkey_small", 3)])
def test_tile_shard_gather_gloo(name, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    frame_ok, stats_ok = q.get(timeout=10)
    assert frame_ok and stats_ok


def test_c_side_partition_covers_every_pixel_once():
    from pyrtx import api
    for (w, h, world) in [(320, 180, 2), (1920, 1080, 8), (301, 173, 3), (32, 32, 4), (64, 32, 5)]:
        seen = np.zeros(w * h, np.int32)
        tiles = ((w + 31) // 32) * ((h + 31) // 32)
        owned = 0
        for r in range(world):
            total, per_rank, own = api.group_layout(w, h, world, r)
            assert total == tiles and per_rank == (tiles + world - 1) // world and own == len(range(r, tiles, world))
            owned += own
            idx = api.group_slot_pixels(w, h, world, r)
            assert idx.size == per_rank * 1024
            assert np.all(idx[own * 1024:] == -1)                      # padding tiles carry no pixel
            np.add.at(seen, idx[idx >= 0], 1)
        assert owned == tiles and np.all(seen == 1), (w, h, world)


def test_c_side_slot_order_is_the_wave_order():
    """slot l of a tile = 8x8 block l >> 6 (4 blocks per tile row), pixel l & 63 in it: 64 consecutive slots = one wave's packet"""
    from pyrtx import api
    idx = api.group_slot_pixels(64, 64, 1, 0).reshape(4, 16, 8, 8)      # tile, block, row, column
    ys, xs = idx // 64, idx % 64
    for t in range(4):
        for b in range(16):
            assert np.array_equal(xs[t, b], np.tile((t % 2) * 32 + (b % 4) * 8 + np.arange(8), (8, 1)))
            assert np.array_equal(ys[t, b], np.tile(((t // 2) * 32 + (b // 4) * 8 + np.arange(8))[:, None], (1, 8)))
