"""Closest-hit hybrid threshold (RTX_PK_DEFER_CLOSEST): per-level launch times of the closest-hit kernel, one frame at a time."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, "..", "cpu-raytracer_amd"))
    import numpy as np
    from pyrtx import api, host
    sc = host.atrium_scene(1920, 1080, 3)
    r = api.Renderer(sc)
    a = r.render(serial=True); b = r.render(serial=True, lane_trace=True)
    same = np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and a["stats"] == b["stats"]
    for _ in range(3): r.render_async(serial=True)
    r.synchronize(); r.enable_timing(True)
    N = 10
    for _ in range(N): r.render_async(serial=True)
    r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
    clo = [ms for n, ms in kt if n == "k_trace_closest"]
    per = [round(float(np.mean(clo[i::4])), 4) for i in range(4)]
    print("T0c=%s grow=%s parity=%s closest per level %s sum %.4f" % (os.environ.get("RTX_PK_DEFER_CLOSEST"), os.environ.get("RTX_PK_GROW"), same, per, sum(per)), flush=True)
else:
    for t in sys.argv[1:]:
        f = t.split(":")
        env = dict(os.environ, RTX_PK_DEFER_CLOSEST=f[0])
        if len(f) > 1: env["RTX_PK_GROW"] = f[1]
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=300)
