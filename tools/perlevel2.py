"""Per-launch kernel durations of one cfg3 frame: packet kernels vs per-lane kernels (serial mode, one launch per level)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
import numpy as np
from pyrtx import api, host
os.environ["RTX_SERIAL_MERGE_ANY"] = "0"
sc = host.atrium_scene(1920, 1080, 3)
r = api.Renderer(sc)
for mode in (dict(), dict(packet_closest=True), dict(lane_trace=True)):
    for _ in range(3): r.render_async(serial=True, **mode)
    r.synchronize(); r.enable_timing(True)
    N = 10
    for _ in range(N): r.render_async(serial=True, **mode)
    r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
    per = len(kt) // N
    print(mode)
    for i in range(per):
        print("  %-16s %.4f ms" % (kt[i][0], float(np.mean([kt[i + per * k][1] for k in range(N)]))))
