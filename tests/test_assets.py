"""SURVEY.md 8f rows 1-2 end to end: asset files -> this repo's own loaders (OBJ/MTL, PNG/TGA, SBVH, TLAS, Scene::update
equivalents in librtx_host.so, pyrtx/assemble.py) -> frame, against frames the REAL reference rendered from the same
asset files and the same scene scripts (tests/golden/*/golden*.npz, oracle/ref_harness/make_goldens.py).

Nothing reference-generated is on the input side here: tests/golden/meshes holds data files of the reference's Data/
directory (.obj/.mtl/.png/.tga), tests/golden/scripts the scene scripts that were also fed to the reference harness."""
import os

import numpy as np
import pytest

import util
from pyrtx import assemble

DATA = os.path.join(util.GOLDEN, "meshes")
MATERIALS_EXTRA = "\nsize 320 180\nmatset mesh:0:0 texture ./Data/LEGOSHLD.tga\n"      # as make_goldens.py appends
CASES = {   # golden name -> (script, extra lines, mip_filter, texture_mode)
    "cube": ("cube.txt", "", 1, 2),
    "monkey_small": ("monkey_small.txt", "", 1, 2),
    "materials_aniso": ("materials.txt", MATERIALS_EXTRA, 1, 2),
    "materials_trilinear": ("materials.txt", MATERIALS_EXTRA, 0, 2),
    "materials_ewa": ("materials.txt", MATERIALS_EXTRA, 2, 2),
    "materials_bilinear": ("materials.txt", MATERIALS_EXTRA, 1, 1),
}


def build(name):
    script, extra, mip_filter, texture_mode = CASES[name]
    text = open(os.path.join(util.GOLDEN, "scripts", script)).read() + extra
    return assemble.scene_from_script(text, DATA, accel="sbvh", mip_filter=mip_filter, texture_mode=texture_mode)


@pytest.mark.parametrize("name", sorted(CASES))
def test_assembled_scene_renders_the_reference_frame_on_the_oracle(name):
    import orc
    sc = build(name)
    _, g = util.load_golden(name)
    out = orc.OracleScene(sc).render(threads=8)
    c = util.compare_to_golden(out, g)
    assert c["stats_equal"] and c["bit_exact"] and c["packed_mismatch"] == 0, c


def test_assembled_scene_equals_the_scene_the_reference_dumped():
    """Piece by piece against the reference's post-Scene::update state (material ids are allotted in a different order:
    the harness's base scene reserves some first, so ids are compared through the records they point to)."""
    sc = build("materials_aniso")
    ref, _ = util.load_golden("materials_aniso")
    assert sc.camera.tobytes() == ref.camera.tobytes() and np.array_equal(sc.sky, ref.sky) and np.array_equal(sc.ambient, ref.ambient)
    for k in ("point_lights", "spot_lights", "dir_lights"):
        assert getattr(sc, k).tobytes() == getattr(ref, k).tobytes(), k
    assert sc.tlas_nodes.tobytes() == ref.tlas_nodes.tobytes() and np.array_equal(sc.tlas_indices, ref.tlas_indices)
    assert len(sc.instances) == len(ref.instances) == 5

    def material(scene, mid):
        m = scene.materials[mid]
        tex = None if m["texture_id"] < 0 else scene.textures[int(m["texture_id"])].texels.tobytes()
        return (m["diffuse"].tobytes(), m["reflection"].tobytes(), m["transmittance"].tobytes(), float(m["index_of_refraction"]), tex)
    for a, b in zip(sc.instances, ref.instances):
        assert a["world"].tobytes() == b["world"].tobytes() and a["world_inv"].tobytes() == b["world_inv"].tobytes()
        x, y = sc.blas[int(a["blas_id"])], ref.blas[int(b["blas_id"])]
        assert x.tri_hot.tobytes() == y.tri_hot.tobytes() and len(x.nodes) == len(y.nodes)
        assert np.array_equal(x.tri_cold["material_id"], y.tri_cold["material_id"])
        for local in np.unique(x.tri_cold["material_id"]):
            assert material(sc, x.material_offset + int(local)) == material(ref, y.material_offset + int(local))
    for k in ("spheres", "planes"):
        a, b = getattr(sc, k), getattr(ref, k)
        assert len(a) == len(b)
        for i in range(len(a)):
            assert material(sc, int(a["material_id"][i])) == material(ref, int(b["material_id"][i]))
            x, y = a[i:i + 1].copy(), b[i:i + 1].copy()
            x["material_id"] = 0; y["material_id"] = 0
            assert x.tobytes() == y.tobytes()


def test_mesh_library_caches_and_bvh_cache_files(tmp_path):
    import shutil
    for f in ("Torus.obj", "Torus.mtl"):
        shutil.copy(os.path.join(DATA, f), tmp_path / f)
    lib = assemble.MeshLibrary("sbvh", use_bvh_cache=True)
    a = lib.load(str(tmp_path / "Torus.obj"))
    assert lib.load(str(tmp_path / "Torus.obj")) == a and len(lib.blas) == 1            # bvh_cache: one BLAS per file name
    assert os.path.exists(tmp_path / "Torus.obj.bvh")                                   # written like BottomLevelBVH::save_to_disk
    lib2 = assemble.MeshLibrary("sbvh", use_bvh_cache=True)
    b = lib2.blas[lib2.load(str(tmp_path / "Torus.obj"))]                               # second run: loaded from the cache file
    assert b.nodes.tobytes() == lib.blas[a].nodes.tobytes() and b.tri_hot.tobytes() == lib.blas[a].tri_hot.tobytes()
    ref = open(os.path.join(util.GOLDEN, "bvh_cache", "Torus.obj.bvh"), "rb").read()   # the file the REAL reference wrote for this mesh
    mine = open(tmp_path / "Torus.obj.bvh", "rb").read()
    assert len(mine) == len(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["materials_aniso", "materials_ewa", "cube"])
def test_assembled_scene_renders_the_reference_frame_on_the_gpu(name):
    from pyrtx import api
    sc = build(name)
    _, g = util.load_golden(name)
    r = api.Renderer(sc)
    out = r.render()
    c = util.compare_to_golden(out, g)
    assert c["stats_equal"] and c["nan_mismatch"] == 0 and c["packed_mismatch"] == 0, c
    assert c["max_abs"] < 1e-4, c                                   # BASELINE.json north_star tolerance ...
    assert c["bit_exact"], c                                        # ... and in fact identical
