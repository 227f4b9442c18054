"""Phase profile of the packet kernels (librtx_hip_prof.so, built with -DRTX_PK_PROF): wave-cycles (s_memtime, 100 MHz-independent shader clock)
spent in packet set-up, the descend loop, leaves, pops.  RTX_HIP_LIB must point at the variant."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
from pyrtx import api, host
from pyrtx.ctypes_structs import RtxStats, RtxWork
bounces = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = host.atrium_scene(1920, 1080, bounces)
r = api.Renderer(sc)
shard = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1/shard of the tiles (tile t with t % shard == 0)
n_tiles = (sc.tile_count + shard - 1) // shard
for _ in range(2): r.render_async(0, shard, n_tiles, serial=True)
r.synchronize()
st, wk = RtxStats(), RtxWork()
r.lib.rtx_get_stats(r.ctx, C.byref(st), C.byref(wk))
raw = [int(getattr(wk, n)) for n in [f[0] for f in RtxWork._fields_[:15]]] + [int(wk.reserved[i]) for i in range(5)]
names = ["setup", "descend", "leaf", "pop", "maxwalk"]
for kind, off in (("any", 0), ("closest", 10)):
    tot = sum(raw[off:off + 5])
    print(kind, "total wave-ticks", tot)
    for i, n in enumerate(names):
        c, k = raw[off + i], raw[off + 5 + i]
        print("   %-8s %14d ticks %5.1f%%  count %9d  ticks/call %8.1f" % (n, c, 100.0 * c / max(tot, 1), k, c / max(k, 1)))
    span = raw[off + 7] - ((1 << 62) - raw[off + 8])
    print("   max packet steps of one walk %d, max iterations of one per-lane phase %d" % (raw[off + 5], raw[off + 9]))
    print("   kernel span %d ticks (first wave start -> last wave end); sum of wave lifetimes %d ticks = %.1f waves busy on average" % (span, raw[off + 6], raw[off + 6] / max(span, 1)))
