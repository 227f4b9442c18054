"""Per-function unit vectors recorded from the REAL reference (tests/golden/unit/unit_vectors.npz, made by
oracle/ref_harness/make_unit_goldens.py): AABB::intersect, Scene::trace_primitives with every RayHit field
(triangle_trace, Sphere::trace, Plane::trace accept branches), Scene::intersect_primitives, the three calc_lighting functions,
Window::plot.  CPU: the oracle's functions reproduce them bit for bit.  GPU (-m gpu): so do the production kernels, through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from pyrtx import scene_io as sio

U = dict(np.load(os.path.join(util.GOLDEN, "unit", "unit_vectors.npz")))
# the lights the harness constructs for the calc_lighting probes (refdump.cpp `unitprobe`)
f32 = np.float32


def _lights():
    pl = np.zeros(1, sio.POINT_LIGHT); pl["colour"] = (20.0, 30.0, 26.0); pl["position"] = 0.0
    sl = np.zeros(1, sio.SPOT_LIGHT); sl["colour"] = (40.0, 40.0, 48.0); sl["position"] = 0.0
    d = np.array([-0.8, -0.55, 0.25], f32); d = d * (f32(1.0) / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], dtype=f32))     # Vector3::normalize, Vector3.h:24-31
    sl["negative_direction"] = -d
    libm = C.CDLL("libm.so.6"); libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    deg = lambda a: f32(a) * f32(3.14159265359) * f32(0.00555555555)                                                              # DEG_TO_RAD, Util.h:14
    sl["inner_cutoff"] = libm.cosf(float(deg(f32(0.5) * f32(50.0)))); sl["outer_cutoff"] = libm.cosf(float(deg(f32(0.5) * f32(70.0))))   # SpotLight.h:13-14
    dl = np.zeros(1, sio.DIR_LIGHT); dl["colour"] = (0.9, 0.8, 0.7)
    v = np.array([0.1, -1.0, 0.1], f32); v = v * (f32(1.0) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=f32))
    dl["negative_direction"] = -v
    return pl, sl, dl


def _check_hits(got, want):
    """All 27 RayHit fields, bit for bit, for the rays that hit; hit flag and distance for the others."""
    hit = want[:, 0] > 0
    assert np.array_equal(got[:, 0] > 0, hit)
    assert util.bit_exact(got[:, 1], want[:, 1])                       # distance (inf on a miss)
    assert util.bit_exact(got[hit], want[hit]), int((got[hit].view(np.uint32) != want[hit].view(np.uint32)).any(axis=1).sum())


def test_oracle_reproduces_reference_unit_vectors():
    import orc
    sc, _ = util.load_golden("materials_aniso")
    o = orc.OracleScene(sc); L = orc.lib()
    rays = U["rays"]
    got = np.zeros((len(rays), 27), np.float32)
    occ3 = np.zeros(len(rays), np.int32); occi = np.zeros(len(rays), np.int32)
    for i, r in enumerate(np.ascontiguousarray(rays[:, :18])):
        L.orc_trace_closest(C.byref(o.struct), r.ctypes.data, got[i].ctypes.data)
        od = np.ascontiguousarray(r[:6])
        occ3[i] = L.orc_trace_any(C.byref(o.struct), od[:3].ctypes.data, od[3:].ctypes.data, 3.0)
        occi[i] = L.orc_trace_any(C.byref(o.struct), od[:3].ctypes.data, od[3:].ctypes.data, 1e30)
    _check_hits(got, rays[:, 18:45])
    assert np.array_equal(occ3 != 0, rays[:, 45] > 0) and np.array_equal(occi != 0, rays[:, 46] > 0)
    # AABB::intersect
    ab = np.ascontiguousarray(U["aabb"])
    node = np.zeros(1, sio.BVH_NODE)
    res = np.zeros(len(ab), np.int32)
    for i, r in enumerate(ab):
        node["aabb_min"] = r[0:3]; node["aabb_max"] = r[3:6]
        res[i] = L.orc_aabb_intersect(node.ctypes.data, np.ascontiguousarray(r[6:9]).ctypes.data, np.ascontiguousarray(r[9:12]).ctypes.data, float(r[12]))
    assert np.array_equal(res != 0, ab[:, 13] > 0)
    # calc_lighting x 3
    pl, sl, dl = _lights()
    li = np.ascontiguousarray(U["light"]); out = np.zeros((len(li), 9), np.float32)
    for i, r in enumerate(li):
        L.orc_calc_lighting(pl.ctypes.data, sl.ctypes.data, dl.ctypes.data, np.ascontiguousarray(r[0:3]).ctypes.data, np.ascontiguousarray(r[3:6]).ctypes.data,
                            np.ascontiguousarray(r[6:9]).ctypes.data, float(r[9]), out[i].ctypes.data)
    assert util.bit_exact(out, li[:, 10:19])
    # Window::plot
    rgb = np.ascontiguousarray(U["plot_rgb"])
    assert np.array_equal(np.array([L.orc_plot(c.ctypes.data) for c in rgb], np.uint32), U["plot_packed"])


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 16, 64])                         # production kernels, RTX_RENDER_LANE_TRACE, RTX_RENDER_PACKET_CLOSEST
def test_gpu_reproduces_reference_unit_vectors(flags):
    from pyrtx import api
    sc, _ = util.load_golden("materials_aniso")
    r = api.Renderer(sc)
    rays = U["rays"]
    _check_hits(r.debug_trace_rays(rays[:, :18], flags), rays[:, 18:45])
    for col, maxd in ((45, 3.0), (46, 1e30)):
        odm = np.concatenate([rays[:, :6], np.full((len(rays), 1), maxd, np.float32)], axis=1)
        assert np.array_equal(r.debug_occluded(odm, flags & 16) != 0, rays[:, col] > 0), maxd
    pl, sl, dl = _lights()
    out9, packed = r.debug_light_plot(pl, sl, dl, U["light"][:, :10], U["plot_rgb"])
    assert util.bit_exact(out9, U["light"][:, 10:19])
    assert np.array_equal(packed, U["plot_packed"])
