#!/usr/bin/env python3
"""Does the host block while submitting frames round-robin to several contexts?  Prints the distribution of render_async call times."""
import os, sys, time
import numpy as np
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
from pyrtx import api, host
import torch
sc = host.atrium_scene(1920, 1080, 3, detail=1)
for nfl in (3, 4):
    rs = [api.Renderer(sc) for _ in range(nfl)]
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    for r, s in zip(rs, streams):
        r.set_stream(s.cuda_stream)
    n = (sc.tile_count + 7) // 8
    for k in range(12):
        rs[k % nfl].render_async(0, 8, n, serial=True)
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for k in range(120):
        a = time.perf_counter()
        rs[k % nfl].render_async(0, 8, n, serial=True)
        ts.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts = np.array(ts) * 1e6
    print(f"frames in flight {nfl}: submit loop {1e3 * (t1 - t0):.1f} ms, drain {1e3 * (t2 - t1):.1f} ms, per-call us: median {np.median(ts):.0f}, p90 {np.percentile(ts, 90):.0f}, max {ts.max():.0f}; calls > 300 us: {(ts > 300).sum()}")
