"""Hybrid any-hit walk: parity + timing of the merged shadow-ray launch for RTX_PK_DEFER thresholds (one process per threshold)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, "..", "cpu-raytracer_amd"))
    import numpy as np
    from pyrtx import api, host
    sc = host.atrium_scene(1920, 1080, 3)
    r = api.Renderer(sc)
    a = r.render(serial=True, packet_closest=True); b = r.render(serial=True, lane_trace=True)
    same = np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and a["stats"] == b["stats"]
    out = []
    for mode in (dict(packet_closest=True), dict(lane_trace=True)):
        for _ in range(3): r.render_async(serial=True, **mode)
        r.synchronize(); r.enable_timing(True)
        for _ in range(10): r.render_async(serial=True, **mode)
        r.synchronize(); kt = r.kernel_times(); r.enable_timing(False)
        g = {}
        for n, ms in kt: g.setdefault(n, []).append(ms)
        out.append({k: round(float(np.sum(v)) / 10, 4) for k, v in g.items()})
    print("T0=%s grow=%s scale=%s lib=%s parity=%s any=%.4f (lane %.4f) closest=%.4f (lane %.4f)" % (os.environ.get("RTX_PK_DEFER"), os.environ.get("RTX_PK_GROW"), os.environ.get("RTX_PK_GRID_SCALE"), os.path.basename(os.environ.get("RTX_HIP_LIB", "default")), same, out[0]["k_trace_any"], out[1]["k_trace_any"], out[0]["k_trace_closest"], out[1]["k_trace_closest"]), flush=True)
else:
    # arguments: T0:grow[:gridscale[:libvariant]] ...
    for spec in sys.argv[1:] or ["8:3"]:
        f = spec.split(":")
        env = dict(os.environ, RTX_PK_DEFER=f[0].split("L")[0], RTX_PK_GROW=f[1] if len(f) > 1 else "3")
        if "L" in f[0]: env["RTX_PK_DEFER_LEAF"] = f[0].split("L")[1]
        if len(f) > 2 and f[2]: env["RTX_PK_GRID_SCALE"] = f[2]
        if len(f) > 3 and f[3]: env["RTX_HIP_LIB"] = os.path.join(HERE, "..", "cpu-raytracer_amd", "csrc", "librtx_hip_%s.so" % f[3])
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=300)
