#!/usr/bin/env python3
"""Host-side cost of one rtx_render_tiles call vs GPU-side floor of a frame with almost no work (1 tile)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3, detail=1)
rs = [api.Renderer(sc) for _ in range(3)]
for r in rs: r.render_async(0, 1, 1); r.synchronize()
for tiles in (1, 255, 2040):
    for nctx in (1, 3):
        n = 200
        for r in rs: r.synchronize()
        t0 = time.perf_counter()
        for i in range(n): rs[i % nctx].render_async(0, 1, tiles, serial=True)
        t1 = time.perf_counter()
        for r in rs: r.synchronize()
        t2 = time.perf_counter()
        print(f"tiles {tiles:5d} contexts {nctx}: host enqueue {1e6*(t1-t0)/n:7.1f} us/frame, total {1e3*(t2-t0)/n:7.3f} ms/frame", flush=True)
