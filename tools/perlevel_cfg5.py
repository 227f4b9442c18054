#!/usr/bin/env python3
"""Per-launch kernel durations of one cfg5 frame (16 instanced tori, spheres, EWA plane; 1920x1080, depth 3), one stream, one frame at a time."""
import os, sys
import numpy as np
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
from pyrtx import api, host
import util
sc, _ = util.load_golden("tori16")
sc.config["width"] = 1920; sc.config["height"] = 1080
fov = float(np.float32(110.0) * np.float32(3.14159265359) * np.float32(0.00555555555))
sc.camera = host.camera_basis(1920, 1080, fov, (-4.694016, 6.446100, -0.572288), (0.268476, 0.423740, -0.133092, 0.854779))
r = api.Renderer(sc)
for _ in range(3):
    r.render_async(serial=True)
r.synchronize()
print(r.stats()[0])
r.enable_timing(True)
reps = 10
for _ in range(reps):
    r.render_async(serial=True)
r.synchronize()
t = r.kernel_times()
per = len(t) // reps
names = [x[0] for x in t[:per]]
ms = np.array([[x[1] for x in t[k * per:(k + 1) * per]] for k in range(reps)]).mean(axis=0)
print("cfg5: " + "  ".join(f"{a.replace('k_trace_', '')[:7]} {b * 1000:.0f}" for a, b in zip(names, ms)) + f"   sum {ms.sum():.3f} ms")
