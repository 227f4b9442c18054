"""librtx_host.so: Scene::update equivalents against data dumped from the REAL reference, BLAS builder
invariants, mip chains, and the cfg3 scene generator (all CPU)."""
import numpy as np
import pytest

import util
from pyrtx import host, scene_io as sio


@pytest.mark.parametrize("name", ["tori16_f1", "cube", "materials_aniso", "monkey"])
def test_instance_update_and_tlas_bit_exact_vs_reference(name):
    sc, _ = util.load_golden(name)
    ipos = sc.extra["IPOS"].view(np.float32).reshape(-1, 7)
    iaab = sc.extra["IAAB"].view(np.float32).reshape(-1, 6)
    for i in range(len(sc.instances)):
        root = sc.blas[sc.instances["blas_id"][i]].nodes[0]
        inst, mn, mx = host.instance_update(ipos[i, :3], ipos[i, 3:], root["aabb_min"], root["aabb_max"], int(sc.instances["blas_id"][i]))
        assert inst["world"].tobytes() == sc.instances["world"][i:i + 1].tobytes()          # Transform::calc_world_matrix
        assert inst["world_inv"].tobytes() == sc.instances["world_inv"][i:i + 1].tobytes()  # Matrix4::invert
        assert np.concatenate([mn, mx]).tobytes() == iaab[i].tobytes()                      # AABB::transform
    nodes, idx = host.Tlas(len(sc.instances)).build(ipos[:, :3], iaab)                      # TopLevelBVH::build_bvh
    assert len(nodes) == len(sc.tlas_nodes) and np.array_equal(idx, sc.tlas_indices)
    for k in range(len(nodes)):
        if k != 1:
            assert nodes[k].tobytes() == sc.tlas_nodes[k].tobytes(), k


def test_plane_update_vs_reference():
    sc, _ = util.load_golden("materials_aniso")
    q = host.axis_angle((0, 1, 0), 0.785398)
    p = host.plane_update((0, -1, 0), q, int(sc.planes["material_id"][0]))
    assert p.tobytes() == sc.planes[:1].tobytes()


def _check_bvh(blas: sio.Blas, n_tris):
    nodes = blas.nodes
    seen = np.zeros(n_tris, np.int32)
    stack = [0]
    hot = blas.tri_hot
    pmin = np.minimum(np.minimum(hot["position_0"], hot["position_0"] + hot["position_edge_1"]), hot["position_0"] + hot["position_edge_2"])
    pmax = np.maximum(np.maximum(hot["position_0"], hot["position_0"] + hot["position_edge_1"]), hot["position_0"] + hot["position_edge_2"])
    while stack:
        k = stack.pop()
        n = nodes[k]
        cnt = int(n["count"]) & 0x3fffffff
        assert np.all(n["aabb_max"] > n["aabb_min"])                   # slab test needs non-empty extents (AABB.h:26-32)
        if cnt > 0:
            f = int(n["left_or_first"])
            seen[f:f + cnt] += 1
            assert np.all(pmin[f:f + cnt] >= n["aabb_min"] - 1e-6) and np.all(pmax[f:f + cnt] <= n["aabb_max"] + 1e-6)
        else:
            l = int(n["left_or_first"])
            assert l >= 2 and (int(n["count"]) >> 30) & 3 in (1, 2, 3)
            for c in (l, l + 1):
                assert np.all(nodes[c]["aabb_min"] >= n["aabb_min"] - 1e-6) and np.all(nodes[c]["aabb_max"] <= n["aabb_max"] + 1e-6)
                stack.append(c)
    assert np.all(seen == 1)


def test_blas_builder_invariants_and_degenerate_inputs():
    rng = np.random.RandomState(1)
    pos = rng.uniform(-5, 5, (3000, 1, 3)).astype(np.float32) + rng.uniform(-0.3, 0.3, (3000, 3, 3)).astype(np.float32)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (3000, 3, 1)); uv = np.zeros((3000, 3, 2), np.float32)
    b = host.build_blas(pos, nrm, uv, np.zeros(3000, np.int32), 1)
    _check_bvh(b, 3000)
    # all triangles identical (coincident centroids) and a single triangle; axis-aligned flat quads
    same = np.tile(pos[:1], (50, 1, 1))
    _check_bvh(host.build_blas(same, nrm[:50], uv[:50], np.zeros(50, np.int32), 1), 50)
    _check_bvh(host.build_blas(pos[:1], nrm[:1], uv[:1], np.zeros(1, np.int32), 1), 1)
    flat = pos.copy(); flat[:, :, 1] = 0.0
    _check_bvh(host.build_blas(flat, nrm, uv, np.zeros(3000, np.int32), 1), 3000)


def test_texture_mips_match_reference_chain():
    sc, _ = util.load_golden("materials_aniso")
    for t in sc.textures:
        w, h = int(t.desc["width"][0]), int(t.desc["height"][0])
        mine = host.texture_with_mips(t.texels[:w * h].reshape(h, w, 3))
        assert mine.desc.tobytes() == t.desc.tobytes()
        assert mine.texels.tobytes() == t.texels.tobytes()              # Texture.cpp:76-117 box filter, bit-exact


def test_atrium_scene_is_deterministic_and_renders_on_the_oracle():
    import orc
    a = host.atrium_scene(160, 90, 3, detail=0)
    b = host.atrium_scene(160, 90, 3, detail=0)
    assert a.blas[0].nodes.tobytes() == b.blas[0].nodes.tobytes() and a.blas[0].tri_hot.tobytes() == b.blas[0].tri_hot.tobytes()
    assert a.blas[0].source_triangle_count == 63992 and a.triangle_refs() >= 63992 and len(a.materials) == 26   # SBVH may duplicate references
    assert host.atrium_scene(160, 90, 3, detail=0, accel="binned").triangle_refs() == 63992
    out = orc.OracleScene(a).render(threads=8)
    assert out["stats"]["primary"] == 160 * 90 and out["stats"]["shadow"] > 0 and out["stats"]["reflection"] > 0 and out["stats"]["refraction"] > 0
    assert not np.isnan(out["rgb"]).any()


@pytest.mark.parametrize("mesh,golden,blas_index", [("Cube", "cube", 0), ("Torus", "tori16_f1", 0)])
def test_reference_bvh_cache_files_load_unchanged(mesh, golden, blas_index, tmp_path):
    """`.obj.bvh` files WRITTEN BY THE REAL REFERENCE (BottomLevelBVH::save_to_disk, committed as data under
    tests/golden/bvh_cache) load through host.load_bvh_cache into exactly the BLAS the reference traversed."""
    import os
    sc, _ = util.load_golden(golden)
    ref = sc.blas[blas_index]
    b = host.load_bvh_cache(os.path.join(util.GOLDEN, "bvh_cache", mesh + ".obj.bvh"), ref.material_offset)
    assert b.nodes.tobytes() == ref.nodes.tobytes()
    assert b.tri_hot.tobytes() == ref.tri_hot.tobytes() and b.tri_cold.tobytes() == ref.tri_cold.tobytes()
    # writer -> reader round trip
    host.save_bvh_cache(str(tmp_path / "x.bvh"), b)
    c = host.load_bvh_cache(str(tmp_path / "x.bvh"), ref.material_offset)
    assert c.nodes.tobytes() == b.nodes.tobytes() and c.tri_hot.tobytes() == b.tri_hot.tobytes() and c.tri_cold.tobytes() == b.tri_cold.tobytes()


def test_camera_update_with_keys_matches_the_reference():
    """Camera::update (Camera.cpp:18-48) with W, D, SPACE, LEFT, UP held for 4 updates of 0.02 s: the camera the REAL reference ended
    up with (golden `camera_keys`, made through the harness's `keys` command) vs rtxh_camera_update + rtxh_camera_basis."""
    sc, g = util.load_golden("camera_keys")
    pos, rot = np.array([0.4, 0.2, -3.0], np.float32), host.axis_angle((0, 1, 0), 0.2)
    for _ in range(4):
        pos, rot = host.camera_update(0.02, ["W", "D", "SPACE", "LEFT", "UP"], pos, rot)
    fov = float(np.float32(110.0) * np.float32(3.14159265359) * np.float32(0.00555555555))
    cam = host.camera_basis(sc.width, sc.height, fov, pos, rot)
    assert cam.tobytes() == sc.camera.tobytes()
    still_p, still_r = host.camera_update(0.02, [], pos, rot)                      # no key: nothing moves
    assert still_p.tobytes() == pos.tobytes() and still_r.tobytes() == rot.tobytes()
    a = host.camera_update(0.02, ["W", "S"], pos, rot)[0]                          # opposite keys do not cancel exactly in fp32 ...
    b = (pos + (host_forward(rot) * np.float32(10.0)) * np.float32(0.02)) - (host_forward(rot) * np.float32(10.0)) * np.float32(0.02)
    assert a.tobytes() == b.astype(np.float32).tobytes()                           # ... they are applied one after the other, like the reference


def host_forward(rot):
    """rotation * (0, 0, 1) through the library (camera basis of a 2x2 view is overkill: use the instance matrix)."""
    inst, _, _ = host.instance_update((0, 0, 0), rot, (-1, -1, -1), (1, 1, 1), 0)
    w = inst["world"][0]                                                            # cells[i + 4 j]: third column = image of z
    return np.array([w[2], w[6], w[10]], np.float32)


def test_sky_probe_file_loader(tmp_path):
    """Sky::Sky (Sky.cpp:8-26): raw float3 texels of a square probe."""
    sky = host.synthetic_sky(64)
    p = tmp_path / "probe.float"
    sky.tofile(str(p))
    assert np.array_equal(host.load_sky(str(p)), sky)
    (tmp_path / "bad.float").write_bytes(sky.tobytes()[:12 * 63 * 64])       # 4032 texels: not a square (the reference asserts)
    with pytest.raises(ValueError):
        host.load_sky(str(tmp_path / "bad.float"))
    with pytest.raises(ValueError):
        host.load_sky(str(tmp_path / "missing.float"))


@pytest.mark.parametrize("mesh", ["Cube", "Torus"])
def test_native_bvh_cache_io(mesh, tmp_path):
    """rtxh_bvh_cache_load / _save (C): files written by the REAL reference load into the same BLAS as through the Python reader,
    and a load -> save round trip reproduces the reference's file byte for byte."""
    import ctypes as C
    import os
    src = os.path.join(util.GOLDEN, "bvh_cache", mesh + ".obj.bvh")
    a, b = host.load_bvh_cache(src), host.load_bvh_cache_native(src)
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.tri_hot.tobytes() == b.tri_hot.tobytes() and a.tri_cold.tobytes() == b.tri_cold.tobytes()
    c = host.RtxhBvhCache()
    assert host.lib().rtxh_bvh_cache_load(src.encode(), C.byref(c)) == 0
    out = str(tmp_path / "copy.bvh")
    assert host.lib().rtxh_bvh_cache_save(out.encode(), C.byref(c)) == 0
    host.lib().rtxh_bvh_cache_free(C.byref(c))
    assert open(out, "rb").read() == open(src, "rb").read()
    raw = open(src, "rb").read()
    for bad in (raw[:len(raw) // 2], raw + b"\\0\\0\\0\\0", b"\\xff\\xff\\xff\\x7f" + raw[4:], b""):       # truncated, trailing bytes, absurd count, empty
        p = tmp_path / "bad.bvh"; p.write_bytes(bad)
        with pytest.raises(ValueError):
            host.load_bvh_cache_native(str(p))


def test_mtl_only_loader_matches_the_obj_loader(tmp_path):
    import ctypes as C
    import os
    obj = os.path.join(util.GOLDEN, "meshes", "Cube.obj")
    full = host.load_obj(obj)
    o = host.RtxhObj()
    assert host.lib().rtxh_mtl_load(obj.encode(), C.byref(o)) == 0
    n = o.mesh.material_count
    mats = np.frombuffer((C.c_char * (n * sio.MATERIAL.itemsize)).from_address(o.materials), sio.MATERIAL, n).copy()
    host.lib().rtxh_obj_free(C.byref(o))
    assert mats.tobytes() == full[4].tobytes()
    assert host.lib().rtxh_mtl_load(str(tmp_path / "none.obj").encode(), C.byref(o)) == 5
