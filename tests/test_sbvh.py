"""the reference's SBVH builder (MESH_ACCELERATOR_SBVH, BVHBuilders.h:48-329, BVHPartitions.h:117-377)
restated in librtx_host.so, against trees built by the REAL reference:
  * the small meshes of the reference's Data/ directory (two of them trigger spatial splits and unsplitting):
    tests/golden/meshes/<mesh>.ref_sbvh.npz, made by oracle/ref_harness/make_mesh_goldens.py;
  * the cfg3 mesh (255 296 triangles, ~3 000 duplicated references): a digest of the `.bvh` cache file the reference
    wrote for it, tests/golden/atrium_d1_ref_sbvh.json, made by oracle/ref_harness/atrium_crosscheck.py.
Nodes the reference allocates but never writes (index 1, the child pair of a node the SAH then turns into a leaf) hold
heap garbage in its dumps, so node arrays are compared over the nodes a traversal can reach, plus their total count."""
import hashlib
import json
import os

import numpy as np
import pytest

import util
from pyrtx import host, obj_export, scene_io as sio

MESHES = ["Cube", "Diamond", "icosphere", "Concave", "Torus", "Monkey", "Rock"]
D = os.path.join(util.GOLDEN, "meshes")


def reachable(nodes):
    cnt = nodes["count"]; lf = nodes["left_or_first"]
    seen = np.zeros(len(nodes), bool); stack = [0]
    while stack:
        k = stack.pop()
        assert not seen[k]
        seen[k] = True
        if (int(cnt[k]) & 0x3fffffff) == 0:
            stack += [int(lf[k]), int(lf[k]) + 1]
    return seen


@pytest.mark.parametrize("mesh", MESHES)
def test_sbvh_equals_the_reference_tree(mesh):
    g = np.load(os.path.join(D, mesh + ".ref_sbvh.npz"))
    pos, nrm, uv, mid, mats, names = host.load_obj(os.path.join(D, mesh + ".obj"))
    b = host.build_blas(pos, nrm, uv, mid, material_offset=0, reference_sbvh=True)
    ref_nodes = g["nodes"]
    assert len(b.nodes) == len(ref_nodes)
    live = reachable(ref_nodes)
    assert np.array_equal(live, reachable(b.nodes))
    assert b.nodes[live].tobytes() == ref_nodes[live].tobytes()
    assert b.tri_hot.tobytes() == g["tri_hot"].tobytes()          # flattened reference order, duplicates included
    assert b.tri_cold.tobytes() == g["tri_cold"].tobytes()
    if mesh in ("Concave", "Monkey"):
        assert len(b.tri_hot) > len(pos)                          # these two do split triangles


def test_sbvh_structure_invariants():
    """Every leaf range lies inside the flattened array, leaves tile it exactly, every source triangle is referenced."""
    pos, nrm, uv, mid, mats, names = host.load_obj(os.path.join(D, "Monkey.obj"))
    b = host.build_blas(pos, nrm, uv, mid, material_offset=0, reference_sbvh=True)
    live = reachable(b.nodes)
    used = np.zeros(len(b.tri_hot), int)
    for k in np.nonzero(live)[0]:
        n = b.nodes[k]
        c = int(n["count"]) & 0x3fffffff
        if c:
            f = int(n["left_or_first"])
            assert 0 <= f and f + c <= len(b.tri_hot)
            used[f:f + c] += 1
        else:
            assert (int(n["count"]) >> 30) & 3 in (1, 2, 3)
    assert np.all(used == 1)                                      # the flattened array is exactly the leaves, back to back
    src = {p.tobytes() for p in pos.reshape(-1, 9)[:, :3].copy()}
    assert {t["position_0"].tobytes() for t in b.tri_hot} == src


def test_sbvh_capacity_is_checked():
    import ctypes as C
    pos, *_ = host.load_obj(os.path.join(D, "Monkey.obj"))
    p = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
    n = len(p)
    nodes = np.zeros(2 * n, sio.BVH_NODE); order = np.zeros(n, np.int32)       # Monkey needs 776 references: n is too few
    nc = C.c_int32(); oc = C.c_int32()
    rc = host.lib().rtxh_blas_build_reference_sbvh(p.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(nc), order.ctypes.data, n, C.byref(oc))
    assert rc == 4                                                # RTX_ERR_LIMIT instead of the reference's silent overrun
    assert host.lib().rtxh_blas_build_reference_sbvh(None, n, nodes.ctypes.data, len(nodes), C.byref(nc), order.ctypes.data, n, C.byref(oc)) == 1


def test_sbvh_cfg3_mesh_equals_the_reference_cache_digest(tmp_path):
    """The reference read the atrium as OBJ text; the same text through this repo's loader and SBVH builder must give
    the bytes of the `.bvh` file the reference wrote (digest committed, the 35 MB file is not)."""
    ref = json.load(open(os.path.join(util.GOLDEN, "atrium_d1_ref_sbvh.json")))
    pos, nrm, uv, mid, nmat = host.atrium_mesh(ref["seed"], ref["detail"])
    path = obj_export.write_obj(str(tmp_path), "atrium", pos, nrm, uv, mid, host.atrium_materials(nmat))
    pos, nrm, uv, mid, mats, names = host.load_obj(path)
    assert len(pos) == ref["source_triangles"]
    b = host.build_blas(pos, nrm, uv, mid, material_offset=0, reference_sbvh=True)
    assert (len(b.nodes), len(b.tri_hot)) == (ref["nodes"], ref["references"])
    assert hashlib.sha256(b.nodes[reachable(b.nodes)].tobytes()).hexdigest() == ref["reachable_nodes_sha256"]
    assert hashlib.sha256(b.tri_hot.tobytes()).hexdigest() == ref["tri_hot_sha256"]
    assert hashlib.sha256(b.tri_cold.tobytes()).hexdigest() == ref["tri_cold_sha256"]


@pytest.mark.parametrize("kind", ["identical", "points", "coplanar", "huge", "nan", "grid", "needles"])
def test_builders_survive_degenerate_meshes(kind):
    """All three builders on meshes the reference itself mishandles (it overruns fixed arrays on reference explosions and on NaN
    keys): a valid tree or an error status, never a crash or a hang (the same matrix ran clean under ASAN + UBSAN)."""
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 17, 300):
        p = rng.normal(size=(n, 3, 3)).astype(np.float32)
        if kind == "identical": p[:] = p[0]
        if kind == "points": p[:, 1] = p[:, 0]; p[:, 2] = p[:, 0]
        if kind == "coplanar": p[:, :, 1] = 0
        if kind == "huge": p *= 1e18
        if kind == "nan": p[::7, 0, 0] = np.nan
        if kind == "grid": p = np.round(p * 2) / 2
        if kind == "needles": p[:, :, 0] *= 1e-6
        z = np.zeros((n, 3, 3), np.float32); uv = np.zeros((n, 3, 2), np.float32); mid = np.zeros(n, np.int32)
        for mode in ("sbvh", "bvh", "binned"):
            try:
                b = host.build_blas(p, z, uv, mid, 0, reference_sbvh=mode == "sbvh", reference_bvh=mode == "bvh")
            except (AssertionError, ValueError):
                assert kind in ("coplanar", "huge", "nan", "needles"), (kind, n, mode)      # these may be refused; the others must build
                continue
            live = reachable(b.nodes)
            covered = sum(int(b.nodes[k]["count"]) & 0x3fffffff for k in np.nonzero(live)[0])
            assert covered == len(b.tri_hot) >= n
