"""Where is the production (packet) path slower than the per-lane kernels?  Every golden scene at 1920x1080, one frame at a time:
frame time and traversal-kernel times per level, default vs RTX_RENDER_LANE_TRACE."""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from pyrtx import api, host
import util
for name in util.GOLDENS:
    if name.endswith("_heat"): continue
    sc, _ = util.load_golden(name)
    sc.config["width"] = 1920; sc.config["height"] = 1080
    try:
        r = api.Renderer(sc)
    except Exception as e:
        print(name, "skipped", e); continue
    row = {}
    for label, mode in (("packet", dict()), ("lane", dict(lane_trace=True))):
        for _ in range(2): r.render_async(serial=True, **mode)
        r.synchronize(); r.enable_timing(True)
        N = 5
        for _ in range(N): r.render_async(serial=True, **mode)
        r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
        g = {}
        for n, ms in kt: g.setdefault(n, []).append(ms)
        row[label] = {k.replace("k_trace_", ""): [round(float(np.mean(v[i::len(v) // N])), 3) for i in range(len(v) // N)] for k, v in g.items() if "trace" in k}
        row[label]["sum"] = round(sum(sum(v) for v in g.values()) / N, 3)
    print(name, "instances", len(sc.instances), row, flush=True)
