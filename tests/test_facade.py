"""The reference-shaped C++ facade (host/rtx_facade.hpp) drives the GPU path and rebuilds the frame state
(instance matrices, TLAS) itself; its output must equal the Python-driven path and the reference goldens."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def fnv1a(packed):
    h = 1469598103934665603
    for b in packed.astype("<u4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.mark.parametrize("group", [False, True])
@pytest.mark.parametrize("name", ["cube", "tori16_f1"])
def test_facade_demo_matches_golden(name, group, tmp_path):
    """group: through rtx::RaytracerGroup (rtx_group_create_local / rtx_group_render_local, real RCCL communicator) with one GPU."""
    d, tag, _ = util.GOLDENS[name]
    raw = tmp_path / "scene.rtxs"
    with gzip.open(os.path.join(util.GOLDEN, d, "scene.rtxs.gz"), "rb") as f:
        raw.write_bytes(f.read())
    sc, g = util.load_golden(name)
    exe = os.path.join(util.REPO, "cpu-raytracer_amd", "host", "facade_demo")
    out = subprocess.run([exe, str(raw), str(sc.width), str(sc.height), str(int(sc.config["bounces"][0])), str(int(sc.config["mip_filter"][0])),
                          str(int(sc.config["texture_mode"][0]))], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, RTX_FACADE_GROUP="1") if group else None)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert [res["primary"], res["shadow"], res["reflection"], res["refraction"]] == g["stats"].tolist()
    assert res["fnv1a"] == fnv1a(g["packed"].reshape(-1))


@pytest.mark.parametrize("name,script,extra,mip_filter,texture_mode", [
    ("materials_aniso", "materials.txt", "\nsize 320 180\nmatset mesh:0:0 texture ./Data/LEGOSHLD.tga\n", 1, 2),
    ("monkey_small", "monkey_small.txt", "", 1, 2),
])
def test_facade_builds_the_scene_from_asset_files(name, script, extra, mip_filter, texture_mode, tmp_path):
    """C++ only: scene script -> Scene::add_mesh (OBJ/MTL loader, SBVH, Texture::load equivalents) -> Scene::update -> frame.
    The packed frame must be the one the REAL reference rendered from the same files (golden)."""
    _, g = util.load_golden(name)
    s = tmp_path / "scene.txt"
    s.write_text(open(os.path.join(util.GOLDEN, "scripts", script)).read() + extra)
    exe = os.path.join(util.REPO, "cpu-raytracer_amd", "host", "facade_demo")
    out = subprocess.run([exe, "--script", str(s), os.path.join(util.GOLDEN, "meshes"), str(mip_filter), str(texture_mode)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert [res["primary"], res["shadow"], res["reflection"], res["refraction"]] == g["stats"].tolist()
    assert res["fnv1a"] == fnv1a(g["packed"].reshape(-1))


def test_facade_bvh_cache_files_like_the_reference(tmp_path):
    """BottomLevelBVH::load keeps `<mesh>.obj.bvh` next to the mesh (BottomLevelBVH.cpp:26-33,53).  With the facade's USE_BVH_CACHE_FILES
    the first run writes the files, the second loads them (OBJLoader::load_mtl for the materials); both runs give the reference's frame,
    and the Torus cache equals the file the REAL reference wrote except for the node the reference never initialises (index 1)."""
    import shutil
    data = tmp_path / "Data"
    shutil.copytree(os.path.join(util.GOLDEN, "meshes"), data)
    _, g = util.load_golden("materials_aniso")
    s = tmp_path / "scene.txt"
    s.write_text(open(os.path.join(util.GOLDEN, "scripts", "materials.txt")).read() + "\nsize 320 180\nmatset mesh:0:0 texture ./Data/LEGOSHLD.tga\n")
    exe = os.path.join(util.REPO, "cpu-raytracer_amd", "host", "facade_demo")
    env = dict(os.environ, RTX_FACADE_BVH_CACHE="1")
    for run in range(2):
        out = subprocess.run([exe, "--script", str(s), str(data), "1", "2"], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 0, out.stderr
        res = json.loads(out.stdout.strip().splitlines()[-1])
        assert res["fnv1a"] == fnv1a(g["packed"].reshape(-1)), run
        assert (data / "Torus.obj.bvh").exists() and (data / "Monkey.obj.bvh").exists()
    mine = bytearray((data / "Torus.obj.bvh").read_bytes())
    ref = bytearray(open(os.path.join(util.GOLDEN, "bvh_cache", "Torus.obj.bvh"), "rb").read())
    n = int(np.frombuffer(bytes(ref[:4]), np.int32)[0])
    node1 = 4 + n * 100 + 4 + 32
    mine[node1:node1 + 32] = ref[node1:node1 + 32]
    assert mine == ref
