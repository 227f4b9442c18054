"""Row a19 end to end: the per-frame Scene::update (Scene.cpp:139-171, animation of SCENE_DYNAMIC :141-155) done with
this repo's host library, frame after frame, must land on the frame state and pixels the REAL reference produced
(goldens `dynamic` = 3 updates, `tori16` = 2 updates, `tori16_f1` = 1 update of delta 1/60 s)."""
import ctypes as C

import numpy as np
import pytest

import util
from pyrtx import host, scene_io as sio

_libm = C.CDLL("libm.so.6")
_libm.sinf.restype = C.c_float; _libm.sinf.argtypes = [C.c_float]
_libm.cosf.restype = C.c_float; _libm.cosf.argtypes = [C.c_float]
f32 = np.float32


def sinf(x): return f32(_libm.sinf(float(f32(x))))
def cosf(x): return f32(_libm.cosf(float(f32(x))))


def quat_axis_angle(axis, angle):           # Quaternion::axis_angle, Quaternion.h:24-34
    half = f32(0.5) * f32(angle)
    s = sinf(half)
    return np.array([f32(axis[0]) * s, f32(axis[1]) * s, f32(axis[2]) * s, cosf(half)], f32)


def quat_mul(l, r):                         # operator*(Quaternion, Quaternion&), Quaternion.h:118-125
    return np.array([l[0] * r[3] + l[3] * r[0] + l[1] * r[2] - l[2] * r[1],
                     l[1] * r[3] + l[3] * r[1] + l[2] * r[0] - l[0] * r[2],
                     l[2] * r[3] + l[3] * r[2] + l[0] * r[1] - l[1] * r[0],
                     l[3] * r[3] - l[0] * r[0] - l[1] * r[1] - l[2] * r[2]], f32)


def quat_nlerp(a, b, t):                    # Quaternion::nlerp, Quaternion.h:107-116
    t = f32(t); omt = f32(1.0) - t
    q = np.array([omt * a[i] + t * b[i] for i in range(4)], f32)
    n = np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], dtype=f32)
    return q * (f32(1.0) / n)


def animate(pos, rot, time, delta):
    """Scene::update body for SCENE_DYNAMIC, Scene.cpp:141-155 (instances 0..5)."""
    delta = f32(delta)
    rot[0] = quat_mul(quat_axis_angle((0, 1, 0), delta), rot[0])
    time = f32(time + delta)
    pos[1][1] = f32(1.0) + f32(2.0) * sinf(time)
    pos[2][0] = pos[2][0] - delta * f32(0.5)
    pos[3] = np.array([6.0, f32(4.0) + f32(2.0) * sinf(time * f32(0.5)), f32(4.0) + f32(2.0) * cosf(time * f32(0.5))], f32)
    rot[3] = quat_mul(quat_axis_angle((0, 1, 0), delta * f32(0.5)), rot[3])
    rot[4] = quat_mul(quat_axis_angle((1, 0, 0), delta), rot[4])
    deg = f32(-90.0) * f32(3.14159265359) * f32(0.00555555555)
    rot[5] = quat_nlerp(np.array([0, 0, 0, 1], f32), quat_axis_angle((1, 0, 0), deg), f32(0.5) + f32(0.5) * sinf(time))
    return time


def initial_state(name):
    if name == "dynamic":                   # Scene.cpp:33-38
        pos = [[0, 1, 0], [4, 2, 0], [0, 3, 4], [6, 4, 4], [0, 5, 8], [-4, 2, 6]]
        rot = [[0, 0, 0, 1]] * 6
    else:                                   # oracle/ref_harness/scenes/tori16.txt
        pos, rot = [], []
        axes = [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
        for k in range(16):
            pos.append([-4.5 + 3.0 * (k % 4), 1.0 + 1.5 * (k // 4), 3.0 + 3.0 * (k // 4)])
            rot.append(quat_axis_angle(axes[k % 3], 0.10 + 0.25 * k))
    return [np.array(p, f32) for p in pos], [np.array(r, f32) for r in rot]


def frame_state(name, frames, delta=0.0166666667):
    sc, g = util.load_golden(name)
    pos, rot = initial_state("dynamic" if name == "dynamic" else "tori16")
    n = len(pos)
    tl = host.Tlas(n)
    time = f32(0.0)
    for _ in range(frames):
        time = animate(pos, rot, time, delta)
        inst = np.zeros(n, sio.INSTANCE); aabbs = np.zeros((n, 6), f32)
        for i in range(n):
            root = sc.blas[sc.instances["blas_id"][i]].nodes[0]
            one, mn, mx = host.instance_update(pos[i], rot[i], root["aabb_min"], root["aabb_max"], int(sc.instances["blas_id"][i]))
            inst[i] = one[0]; aabbs[i, :3] = mn; aabbs[i, 3:] = mx
        nodes, idx = tl.build(np.array(pos, f32), aabbs)
    return sc, g, inst, nodes, idx


@pytest.mark.parametrize("name,frames", [("dynamic", 3), ("tori16", 2), ("tori16_f1", 1)])
def test_host_update_sequence_reproduces_reference_frame_state(name, frames):
    sc, g, inst, nodes, idx = frame_state(name, frames)
    assert inst["world"].tobytes() == sc.instances["world"].tobytes()
    assert inst["world_inv"].tobytes() == sc.instances["world_inv"].tobytes()
    assert np.array_equal(idx, sc.tlas_indices) and len(nodes) == len(sc.tlas_nodes)
    for k in range(len(nodes)):
        if k != 1:
            assert nodes[k].tobytes() == sc.tlas_nodes[k].tobytes(), k


@pytest.mark.gpu
@pytest.mark.parametrize("name,frames", [("dynamic", 3), ("tori16", 2)])
def test_gpu_renders_rebuilt_frames(name, frames):
    """set_frame with the host-rebuilt TLAS / matrices every frame, then render: last frame equals the reference golden."""
    from pyrtx import api
    sc, g, inst, nodes, idx = frame_state(name, 1)
    r = api.Renderer(sc)
    for f in range(1, frames + 1):
        sc_f, _, inst, nodes, idx = frame_state(name, f)
        sc.instances, sc.tlas_nodes, sc.tlas_indices = inst, nodes, idx
        r.set_frame(sc)
        out = r.render()
    cmp = util.compare_to_golden(out, g)
    assert cmp["stats_equal"] and cmp["max_abs"] < 1e-4 and cmp["packed_mismatch"] == 0, cmp
    assert cmp["max_abs"] == 0.0, cmp


@pytest.mark.parametrize("name,frames", [("dynamic", 3), ("tori16", 2), ("tori16_f1", 1)])
def test_native_scene_update_reproduces_reference_frame_state(name, frames):
    """The same sequence through the two native calls a frame loop makes (rtxh_scene_dynamic_animate + rtxh_scene_update)."""
    sc, _ = util.load_golden(name)
    pos, rot = initial_state("dynamic" if name == "dynamic" else "tori16")
    dyn = host.DynamicScene(sc, np.array(pos, f32), np.array(rot, f32))
    for _ in range(frames):
        dyn.animate_dynamic(0.0166666667)
        inst, nodes, idx = dyn.update()
    assert inst["world"].tobytes() == sc.instances["world"].tobytes() and inst["world_inv"].tobytes() == sc.instances["world_inv"].tobytes()
    assert np.array_equal(inst["blas_id"], sc.instances["blas_id"])
    assert np.array_equal(idx, sc.tlas_indices) and len(nodes) == len(sc.tlas_nodes)
    for k in range(len(nodes)):
        if k != 1:
            assert nodes[k].tobytes() == sc.tlas_nodes[k].tobytes(), k
