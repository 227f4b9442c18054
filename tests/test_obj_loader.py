"""the OBJ / MTL loader and the reference-topology BLAS builder of librtx_host.so against
BottomLevelBVHs built by the REAL reference (MESH_ACCELERATOR_BVH build, oracle/ref_harness/make_mesh_goldens.py).
The .obj / .mtl files under tests/golden/meshes are data files of the reference's Data/ directory."""
import os

import numpy as np
import pytest

import util
from pyrtx import host, scene_io as sio

MESHES = ["Cube", "Diamond", "icosphere", "Concave", "Torus", "Monkey", "Rock"]
D = os.path.join(util.GOLDEN, "meshes")


@pytest.mark.parametrize("mesh", MESHES)
def test_obj_loader_and_reference_bvh_are_bit_exact(mesh):
    g = np.load(os.path.join(D, mesh + ".ref_bvh.npz"))
    pos, nrm, uv, mid, mats, names = host.load_obj(os.path.join(D, mesh + ".obj"))
    assert len(pos) == int(g["source_triangles"])                              # triangulation (quads, n-gons) agrees
    b = host.build_blas(pos, nrm, uv, mid, material_offset=0, reference_bvh=True)
    ref_nodes = g["nodes"].copy()
    assert len(b.nodes) == len(ref_nodes)
    for k in range(len(ref_nodes)):
        if k != 1:
            assert b.nodes[k].tobytes() == ref_nodes[k].tobytes(), (mesh, k)
    assert b.tri_hot.tobytes() == g["tri_hot"].tobytes()                        # positions + leaf order
    assert b.tri_cold.tobytes() == g["tri_cold"].tobytes()                      # uv (v flipped), normals, material ids
    # materials: Kd / Ks / Kt / Ni as OBJLoader.cpp:8-41 (texture pointers are not comparable)
    ref_m = g["materials"]
    assert len(mats) == len(ref_m)
    for f in ("diffuse", "reflection", "transmittance", "index_of_refraction"):
        assert mats[f].tobytes() == ref_m[f].tobytes(), (mesh, f)
    assert [n is not None for n in names] == [int(t) >= 0 for t in ref_m["texture_id"]]


def test_obj_number_parser_quirks():
    """The vendored parser is not strtod: 8+ fractional digits go through pow(10,-k); exponents through 5^e * 2^e."""
    import tempfile
    src = "v 0.123456789 -1.5e-3 7\nv .25 +3.0E2 -0\nv 1e0 2 3\nvn 0 1 0\nf 1//1 2//1 3//1\n"
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.obj")
        open(p, "w").write(src)
        pos, nrm, uv, mid, mats, names = host.load_obj(p)
    assert pos.shape == (1, 3, 3) and np.all(uv == 0) and mats["diffuse"].tolist() == [[1.0, 0.0, 1.0]]   # magenta default
    assert pos[0, 0].tolist() == [np.float32(0.123456789), np.float32(-0.0015), 7.0]
    assert pos[0, 1].tolist() == [0.25, 300.0, 0.0]


@pytest.mark.gpu
def test_scene_built_from_obj_renders_like_the_oracle():
    """End to end without any reference-built data: OBJ -> own BLAS builder -> scene -> GPU vs oracle."""
    import orc
    from pyrtx import api
    pos, nrm, uv, mid, mats, names = host.load_obj(os.path.join(D, "Monkey.obj"))
    sc = sio.Scene()
    sc.config = host.make_config(320, 180, 3)
    m = np.zeros(1 + len(mats), sio.MATERIAL); m["texture_id"] = -1; m["index_of_refraction"] = 1.0
    m[1:] = mats; m["texture_id"][1:] = -1
    m["reflection"][1] = (0.3, 0.3, 0.3)
    sc.materials = m
    sc.blas = [host.build_blas(pos, nrm, uv, mid, material_offset=1)]
    root = sc.blas[0].nodes[0]
    inst, mn, mx = host.instance_update((0, 0, 0), host.axis_angle((0, 1, 0), 0.4), root["aabb_min"], root["aabb_max"], 0)
    sc.instances = inst
    sc.tlas_nodes, sc.tlas_indices = host.Tlas(1).build(np.zeros((1, 3), np.float32), np.concatenate([mn, mx])[None])
    sc.sky = host.synthetic_sky()
    sc.dir_lights = np.zeros(1, sio.DIR_LIGHT); sc.dir_lights["colour"] = 0.9; sc.dir_lights["negative_direction"] = (0.0, 1.0, 0.0)
    sc.camera = host.camera_basis(320, 180, 1.9198622, (0, 0, -2.2), (0, 0, 0, 1))
    out = api.Renderer(sc).render()
    ref = orc.OracleScene(sc).render(threads=8)
    assert out["stats"] == ref["stats"] and out["stats"]["reflection"] > 0
    assert np.array_equal(out["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))


def test_obj_loader_rejects_out_of_range_indices_and_survives_garbage(tmp_path):
    """tinyobjloader does not range-check face indices (the reference then reads past its arrays); here such a file is an error.
    Random damage to OBJ / MTL text must load or be rejected, never crash (the same loop ran clean under ASAN + UBSAN)."""
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(RuntimeError):
        host.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/1 2/5 3/1\n")
    with pytest.raises(RuntimeError):
        host.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\n")                      # relative indices are fine
    assert host.load_obj(str(p))[0].shape == (1, 3, 3)
    rng = np.random.default_rng(9)
    raw = open(os.path.join(D, "Concave.obj"), "rb").read()
    mtl = open(os.path.join(D, "Concave.mtl"), "rb").read()
    loaded = rejected = 0
    for k in range(60):
        b, m = bytearray(raw), bytearray(mtl)
        tgt = b if k % 3 else m
        if k % 4 == 0:
            for _ in range(int(rng.integers(1, 8))):
                tgt[int(rng.integers(0, len(tgt)))] = int(rng.integers(0, 256))
        elif k % 4 == 1:
            del tgt[int(rng.integers(0, len(tgt))):]
        elif k % 4 == 2:
            i = int(rng.integers(0, len(tgt))); tgt[i:i] = b"f 1/2/3 -5/-6/-7 99999999/1/1 4//4\nv 1e999 -1e-999 nan\nvt\nvn 1\nusemtl nothing\nmtllib\n"
        else:
            i = int(rng.integers(0, len(tgt))); tgt[i:i] = bytes(rng.integers(0, 256, 30).astype(np.uint8))
        (tmp_path / "m.obj").write_bytes(bytes(b).replace(b"Concave.mtl", b"m.mtl")); (tmp_path / "m.mtl").write_bytes(bytes(m))
        try:
            pos, nrm, uv, mid, mats, names = host.load_obj(str(tmp_path / "m.obj")); loaded += 1
            assert len(pos) == len(mid) and mid.max() < len(mats)
        except RuntimeError:
            rejected += 1
    assert loaded > 10
