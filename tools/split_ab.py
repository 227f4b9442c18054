"""Split shadow-ray walk (RTX_PK_SPLIT = items per chunk): parity vs the per-lane kernels and the times of the packet launch + the item launch."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, "..", "cpu-raytracer_amd"))
    import numpy as np
    from pyrtx import api, host
    sc = host.atrium_scene(1920, 1080, 3)
    r = api.Renderer(sc)
    a = r.render(serial=True); b = r.render(serial=True, lane_trace=True); c = r.render()
    same = np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and a["stats"] == b["stats"] and np.array_equal(c["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    for _ in range(3): r.render_async(serial=True)
    r.synchronize(); r.enable_timing(True)
    N = 10
    for _ in range(N): r.render_async(serial=True)
    r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
    g = {}
    for n, ms in kt: g.setdefault(n, []).append(ms)
    print("split=%s T0=%s parity=%s" % (os.environ.get("RTX_PK_SPLIT"), os.environ.get("RTX_PK_DEFER"), same), {k: round(float(np.sum(v)) / N, 4) for k, v in g.items()}, "frame %.3f" % (sum(sum(v) for v in g.values()) / N), flush=True)
else:
    for spec in sys.argv[1:] or ["0", "16384"]:
        f = spec.split(":")
        env = dict(os.environ, RTX_PK_SPLIT=f[0])
        if len(f) > 1: env["RTX_PK_DEFER"] = f[1]
        if len(f) > 2 and f[2]: env["RTX_PK_DEFER_LEAF"] = f[2]
        if len(f) > 3 and f[3]: env["RTX_ITEM_GRID"] = f[3]
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=300)
