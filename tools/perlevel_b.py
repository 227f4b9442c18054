#!/usr/bin/env python3
"""Per-launch kernel durations of one cfg3 frame at a given bounce count (argv[1], default 0), one stream, one frame at a time."""
import os, sys
import numpy as np
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
from pyrtx import api, host
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = host.atrium_scene(1920, 1080, b, detail=1)
r = api.Renderer(sc)
for _ in range(3):
    r.render_async(serial=True)
r.synchronize()
r.enable_timing(True)
reps = 10
for _ in range(reps):
    r.render_async(serial=True)
r.synchronize()
t = r.kernel_times()
per = len(t) // reps
names = [x[0] for x in t[:per]]
ms = np.array([[x[1] for x in t[k * per:(k + 1) * per]] for k in range(reps)]).mean(axis=0)
print(f"bounces {b} RTX_FUSE_SHADE={os.environ.get('RTX_FUSE_SHADE', '1')}: " + "  ".join(f"{a.replace('k_trace_', '')[:13]} {x * 1000:.0f}" for a, x in zip(names, ms)) + f"   sum {ms.sum():.3f} ms")
