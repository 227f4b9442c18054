"""Per-level launch times, packet kernels vs per-lane kernels, for the atrium (cfg3) and the 16-tori scene (cfg5); RTX_SERIAL_MERGE_ANY=0 so that
shadow rays are launched per level too."""
import os, sys
os.environ["RTX_SERIAL_MERGE_ANY"] = "0"
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from pyrtx import api, host
import util
def scenes():
    yield "atrium", host.atrium_scene(1920, 1080, 3)
    sc, _ = util.load_golden("tori16")
    sc.config["width"] = 1920; sc.config["height"] = 1080
    fov = float(np.float32(110.0) * np.float32(3.14159265359) * np.float32(0.00555555555))
    sc.camera = host.camera_basis(1920, 1080, fov, (-4.694016, 6.446100, -0.572288), (0.268476, 0.423740, -0.133092, 0.854779))
    yield "cfg5", sc
for name, sc in scenes():
    r = api.Renderer(sc)
    for mode in (dict(), dict(lane_trace=True)):
        for _ in range(3): r.render_async(serial=True, **mode)
        r.synchronize(); r.enable_timing(True)
        N = 10
        for _ in range(N): r.render_async(serial=True, **mode)
        r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
        g = {}
        for n, ms in kt: g.setdefault(n, []).append(ms)
        print(name, mode, {k: [round(float(np.mean(v[i::len(v) // N])), 4) for i in range(len(v) // N)] for k, v in g.items() if "trace" in k}, flush=True)
