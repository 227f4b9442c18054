import os, sys, ctypes as C
sys.path.insert(0, "/root/repo/cpu-raytracer_amd")
from pyrtx import api, host
from pyrtx.ctypes_structs import RtxStats, RtxWork
sc = host.atrium_scene(1920, 1080, 3)
r = api.Renderer(sc)
r.render_async(serial=True); r.synchronize()
st, wk = RtxStats(), RtxWork()
r.lib.rtx_get_stats(r.ctx, C.byref(st), C.byref(wk))
v = [int(wk.reserved[i]) for i in range(5)]
print("live items %d  wave-iterations %d  busy lane-iterations %d (lane occupancy %.2f, %.2f iterations per item)  most batches of one wave %d  most iterations of one wave %d  avg iterations per wave %.1f" % (v[0], v[1], v[2], v[2] / max(v[1], 1) / 64, v[2] / max(v[0], 1), v[3], v[4], v[1] / (256*4*7)))
