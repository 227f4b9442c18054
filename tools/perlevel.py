#!/usr/bin/env python3
"""Per-launch kernel durations of one cfg3 frame, one stream, one frame at a time (HIP events): which launches are floors?"""
import os, sys
import numpy as np
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3, detail=1)
if os.environ.get("PERLEVEL_CAM_Y"):      # A/B: the camera off the tree's box planes (its default height 5.5 IS a plane of the atrium's tree: the middle row of pixels takes the reference-form walker)
    sc.camera = host.camera_basis(1920, 1080, float(np.deg2rad(110.0)), (host.ATRIUM_CAMERA[0][0], float(os.environ["PERLEVEL_CAM_Y"]), host.ATRIUM_CAMERA[0][2]), host.axis_angle(*host.ATRIUM_CAMERA[1]))
r = api.Renderer(sc)
for shard in (1, 8):
    n = (sc.tile_count + shard - 1) // shard
    for _ in range(3):
        r.render_async(0, shard, n, serial=True)
    r.synchronize()
    r.enable_timing(True)
    reps = 10
    for _ in range(reps):
        r.render_async(0, shard, n, serial=True)
    r.synchronize()
    t = r.kernel_times()
    r.enable_timing(False)
    per = len(t) // reps
    names = [x[0] for x in t[:per]]
    ms = np.array([[x[1] for x in t[k * per:(k + 1) * per]] for k in range(reps)]).mean(axis=0)
    print(f"tile shard 1/{shard}: " + "  ".join(f"{a.replace('k_trace_', '')[:7]} {b * 1000:.0f}" for a, b in zip(names, ms)) + f"   sum {ms.sum():.3f} ms")
