import sys, time, json
sys.path.insert(0, "/root/repo/cpu-raytracer_amd"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import util, orc
from pyrtx import api, host
ok = True
for name in ["cube", "monkey_small", "materials_aniso", "dynamic", "tori16", "materials_ewa"]:
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    a = r.render()
    b = r.render(lane_trace=True)
    c = util.compare_to_golden(a, g)
    same = np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and a["stats"] == b["stats"]
    print(name, "packet==lane", same, "golden", c["bit_exact"], c["stats_equal"], c["packed_mismatch"], flush=True)
    ok &= same and c["bit_exact"]
sc = host.atrium_scene(1920, 1080, 3)
r = api.Renderer(sc)
a = r.render(); b = r.render(lane_trace=True); s = r.render(serial=True)
print("atrium packet==lane", np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)), "serial", np.array_equal(a["rgb"].view(np.uint32), s["rgb"].view(np.uint32)), a["stats"], flush=True)
st = r.render(packet_stats=True, serial=True)
print({k: v for k, v in st["work"].items() if k.startswith("pk_")})
for mode in [dict(), dict(lane_trace=True)]:
    for serial in (False, True):
        r.enable_timing(True)
        for _ in range(5): r.render_async(serial=serial, **mode)
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): r.render_async(serial=serial, **mode)
        r.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        kt = r.kernel_times(); r.enable_timing(False)
        g = {}
        for n, ms in kt: g.setdefault(n, []).append(ms)
        print(mode, "serial" if serial else "overlap", "ms/frame %.3f" % dt, {k: (len(v) // 25, round(float(np.sum(v)) / 25, 4)) for k, v in g.items()}, flush=True)
print("OK" if ok else "FAIL")
