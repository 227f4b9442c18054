"""N > 1 path on CPU: world-size-2 gloo run of the tile sharding + gather used by bench.py
(pyrtx/shard.py).  Each rank renders its tile shard with the oracle (test infrastructure standing in
for the GPU), gathers packed pixels to rank 0, and rank 0 checks the assembled frame."""
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
    from pyrtx import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, g = util.load_golden(name)
    first, stride, count = shard.shard_args(sc.tile_count, rank, world)
    out = orc.OracleScene(sc).render(first_tile=first, tile_stride=stride, tile_count=count, threads=2)
    packed = torch.from_numpy(out["packed"].reshape(-1).view(np.int32).copy())
    gather = shard.TileGather(sc.width, sc.height, rank, world, "cpu")
    final = gather(packed)
    stats = torch.tensor([out["stats"][k] for k in ("primary", "shadow", "reflection", "refraction")], dtype=torch.int64)
    dist.all_reduce(stats)
    if rank == 0:
        ok = np.array_equal(final.numpy().view(np.uint32).reshape(sc.height, sc.width), g["packed"])
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


def test_tile_partition_covers_every_pixel_once():
    from pyrtx import shard
    for (w, h, world) in [(320, 180, 2), (1920, 1080, 8), (301, 173, 3), (32, 32, 4)]:
        seen = np.zeros(w * h, np.int32)
        for r in range(world):
            idx = shard.tile_pixel_indices(w, h, r, world)
            np.add.at(seen, idx[idx >= 0], 1)
        assert np.all(seen == 1), (w, h, world)
