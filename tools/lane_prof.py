"""Where do the microseconds of a dependent step go in the closest-hit per-lane phase?  (librtx_hip_laneprof.so = make variant NAME=laneprof
TUNE=-DRTX_LANE_PROF; RTX_HIP_LIB must point at it; RTX_LANE_PROF_LEVEL picks the level, default 2.)  s_memtime ticks are 10 ns."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
from pyrtx import api, host
from pyrtx.ctypes_structs import RtxStats, RtxWork
sc = host.atrium_scene(1920, 1080, int(os.environ.get("RTX_PROF_BOUNCES", "3")))
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = api.Renderer(sc)
n = (sc.tile_count + shard - 1) // shard
for _ in range(2):
    r.render_async(0, shard, n, serial=True)
r.synchronize()
st, wk = RtxStats(), RtxWork()
r.lib.rtx_get_stats(r.ctx, C.byref(st), C.byref(wk))
raw = [int(getattr(wk, f[0])) for f in RtxWork._fields_[:15]] + [int(wk.reserved[i]) for i in range(5)]
tick = 0.01 * raw[16] / max(raw[14], 1)        # us per s_memtime tick, calibrated on the constant 100 MHz clock (s_memrealtime)
print(f"s_memtime tick = {tick * 1000:.3f} ns")
lvl = os.environ.get("RTX_LANE_PROF_LEVEL", "2")
print(f"level {lvl}, tile shard 1/{shard}: {raw[15]} waves, {raw[12]} per-lane phases")
print(f"  longest wave lifetime {raw[13] * tick:.1f} us, mean {raw[14] / max(raw[15], 1) * tick:.1f} us")
print(f"  longest phase {raw[10] * tick:.1f} us with at most {raw[11]} iterations; mean phase {raw[0] / max(raw[12], 1) * tick:.1f} us, {raw[1] / max(raw[12], 1):.1f} iterations")
print(f"  per iteration: {raw[0] / max(raw[1], 1) * tick:.3f} us;  longest phase / its iterations ~ {raw[10] / max(raw[11], 1) * tick:.3f} us")
it = max(raw[1], 1)
print(f"  lanes still walking per iteration: {raw[6] / it:.1f} of 64 ({raw[6] / it / 64:.2f}); lane-iterations per phase {raw[6] / max(raw[12], 1):.0f}")
print(f"  iterations with an inner step {raw[2] / it:.2f}, with a leaf {raw[3] / it:.2f}, with a stack pop {raw[4] / it:.2f}, with a work-list read {raw[5] / it:.2f}")
print(f"  ticks: child-pair fetch wait {raw[7] / max(raw[0], 1):.2f} of the phase ({raw[7] / max(raw[2], 1) * tick:.3f} us per inner step), "
      f"leaf (fetch + test, serial per triangle) {raw[8] / max(raw[0], 1):.2f} ({raw[8] / max(raw[3], 1) * tick:.3f} us per leaf step), "
      f"pop / work list {raw[9] / max(raw[0], 1):.2f} ({raw[9] / max(raw[4] + raw[5], 1) * tick:.3f} us per pop)")
if raw[19]:
    print(f"  shared (asm) walks: {raw[19]}, {raw[18] / raw[19]:.1f} node visits and {raw[17] / raw[19] * tick:.2f} us each ({raw[17] / max(raw[18], 1) * tick:.3f} us per node visit), {raw[17] / max(raw[14], 1):.2f} of the waves' lifetime")
