"""The drop-in binding EXECUTED.

oracle/_ref/refbind_* = the reference's own translation units (Scene, OBJLoader, BottomLevelBVH, Raytracer, ... compiled from
/root/reference in the build container, SIMD_LANE_SIZE 1) + cpu-raytracer_amd/host/reference_binding/GpuBackend.cpp (the file a maintainer
compiles INSTEAD of WorkerThread.cpp) + librtx_hip.so.  In one process it runs Main.cpp:51-57's sequence on the reference's own Scene and
Window objects — scene.update(delta); WorkerThreads::wake_up_worker_threads(tiles); WorkerThreads::wait_on_worker_threads() — and, on the
same objects, the reference's own Raytracer::render_tile loop (WorkerThread.cpp:53-65); Window::frame_buffer after the two must be equal
in every pixel, frame by frame, and WorkerThreads::sum_performance_stats() must return the reference's ray counts.

Only the binaries travel to the GPU box (oracle/_ref/ is git-ignored, not gpurun-ignored); the meshes / textures they load are the data
files under tests/golden/meshes, the sky probe is the synthetic one (the reference's own probes are absent from its repository)."""
import json
import os
import shutil
import subprocess

import pytest

from util import REPO

pytestmark = pytest.mark.gpu

MESHES = os.path.join(REPO, "tests", "golden", "meshes")
SCENES = os.path.join(REPO, "oracle", "ref_harness", "scenes")


def _stage(work):
    from pyrtx import host
    data = os.path.join(work, "Data")
    os.makedirs(os.path.join(data, "Sky_Probes")); os.makedirs(os.path.join(data, "sponza"))
    for f in os.listdir(MESHES):
        if f.rsplit(".", 1)[-1] in ("obj", "mtl", "png", "tga"):
            shutil.copy(os.path.join(MESHES, f), os.path.join(data, f))
    # SCENE_SPONZA's constructor opens three meshes before the script replaces them (Scene.cpp:77-79): a small one under each name
    shutil.copy(os.path.join(MESHES, "Diamond.obj"), os.path.join(data, "sponza", "sponza.obj"))
    shutil.copy(os.path.join(MESHES, "Diamond.mtl"), os.path.join(data, "sponza", "Diamond.mtl"))
    shutil.copy(os.path.join(MESHES, "Diamond.obj"), os.path.join(data, "Magnifier.obj"))
    host.synthetic_sky().tofile(os.path.join(data, "Sky_Probes", "rnl_probe.float"))


CASES = [("cube", "s0_m1_b0_t2", "", 1),
         ("materials", "s0_m1_b3_t2", "size 320 180\nmatset mesh:0:0 texture ./Data/LEGOSHLD.tga\n", 1),
         ("dynamic", "s1_m2_b3_t2", "size 320 180\n", 3),          # SCENE_DYNAMIC as shipped: animated instances, TLAS rebuilt by Scene::update every frame
         ("tori16", "s1_m2_b3_t2", "size 320 180\n", 2)]


@pytest.mark.parametrize("scene,variant,extra,frames", CASES, ids=[c[0] for c in CASES])
def test_binding_frame_buffer_equals_the_reference_render_tile_loop(tmp_path, scene, variant, extra, frames):
    exe = os.path.join(REPO, "oracle", "_ref", "refbind_" + variant)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs /root/reference: python -c 'import __graft_entry__ as g; g.build()' in the build container)")
    work = str(tmp_path)
    _stage(work)
    script = os.path.join(work, "scene.txt")
    with open(script, "w") as f:
        f.write(open(os.path.join(SCENES, scene + ".txt")).read() + "\n" + extra)
    p = subprocess.run([exe, script, os.path.join(work, "out"), "--gpu-binding", str(frames)], cwd=work, capture_output=True, text=True, timeout=600)
    assert os.path.exists(os.path.join(work, "out", "binding.json")), (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    j = json.load(open(os.path.join(work, "out", "binding.json")))
    assert j["frames"] == frames and j["simd_lane_size"] == 1
    assert j["mismatched_pixels"] == [0] * frames and j["stats_equal"] == [True] * frames and j["identical"] is True, j
    assert j["reference_rays"] == j["binding_rays"] and j["reference_rays"][0] == j["width"] * j["height"]
    assert p.returncode == 0
