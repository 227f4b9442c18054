import os, sys, time
sys.path.insert(0, "/root/repo/cpu-raytracer_amd")
import numpy as np
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3)
r = api.Renderer(sc)
for mode in (dict(), dict(serial=True)):
    for _ in range(5): r.render_async(**mode)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): r.render_async(**mode)
    r.synchronize()
    print("grid_any=%s grid_closest=%s %s ms/frame %.3f" % (os.environ.get("RTX_PK_GRID_ANY"), os.environ.get("RTX_PK_GRID_CLOSEST"), "serial" if mode else "two-stream", (time.perf_counter() - t0) / 30 * 1e3), flush=True)
