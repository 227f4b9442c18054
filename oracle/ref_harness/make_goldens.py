#!/usr/bin/env python3
"""Generate tests/golden/* by running the REAL reference (oracle/_ref/refdump_*) on scene scripts.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference for its Data/ files
and for building oracle/_ref).  What is committed: this script, the scene scripts, and its small
outputs (scene files in this repo's own format + golden pixels).  No reference source or binary is
committed.

Staging: the reference opens files relative to ./Data (Util.h:6), writes `.bvh` caches next to the
OBJ files (BottomLevelBVH.cpp:26,53) and needs a sky probe that is missing from the mount
(.MISSING_LARGE_BLOBS), so a scratch working directory is populated with COPIES of the reference's
data files plus a synthetic 64x64 sky probe, and refdump is run from there.
"""
import gzip
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
WORK = os.environ.get("RTX_REF_WORK", "/tmp/rtx_ref_work")
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
from pyrtx import scene_io as sio  # noqa: E402


def synthetic_sky(n=64):
    """Seeded stand-in for Data/Sky_Probes/rnl_probe.float (absent from the mount)."""
    y, x = np.mgrid[0:n, 0:n]
    sky = np.zeros((n, n, 3), np.float32)
    sky[..., 0] = 0.5 + 0.5 * x / n
    sky[..., 1] = 0.6 + 0.2 * (((x // 8) + (y // 8)) & 1)
    sky[..., 2] = 0.5 + 0.5 * y / n
    return sky


def stage():
    data = os.path.join(WORK, "Data")
    if os.path.isdir(data):
        return
    os.makedirs(os.path.join(data, "Sky_Probes"))
    os.makedirs(os.path.join(data, "sponza"))
    for f in os.listdir(os.path.join(REF, "Data")):
        src = os.path.join(REF, "Data", f)
        if os.path.isfile(src) and f.split(".")[-1] in ("obj", "mtl", "png", "tga"):
            shutil.copy(src, os.path.join(data, f))
    shutil.copy(os.path.join(data, "floor.png"), os.path.join(data, "Floor.png"))      # Scene.cpp:23 asks for Floor.png
    # SCENE_SPONZA's ctor opens sponza/sponza.obj (absent): stage a small mesh under that name; scripts replace it
    shutil.copy(os.path.join(REF, "Data", "Diamond.obj"), os.path.join(data, "sponza", "sponza.obj"))
    shutil.copy(os.path.join(REF, "Data", "Diamond.mtl"), os.path.join(data, "sponza", "Diamond.mtl"))
    synthetic_sky().tofile(os.path.join(data, "Sky_Probes", "rnl_probe.float"))


def run_ref(variant, script_path, name):
    exe = os.path.join(REPO, "oracle", "_ref", "refdump_" + variant)
    if not os.path.exists(exe):
        parts = variant.split("_")
        s, m, b, t = [x[1:] for x in parts[:4]]
        aniso = [x[1:] + ".0f" for x in parts[4:] if x[0] == "a" and x[1:].isdigit()]
        subprocess.check_call(["make", "-C", HERE, "variant", f"SCENE={s}", f"MIP={m}", f"BOUNCES={b}", f"TEXMODE={t}",
                               "ACCEL=" + ("0" if "bvh" in parts[4:] else "1"), "HEAT=" + ("true" if "heat" in parts[4:] else "false"),
                               "TRAVERSAL=" + ("0" if "naive" in parts[4:] else "1"), "ANISO=" + (aniso[0] if aniso else "8.0f")])
    out = os.path.join(WORK, "out_" + name)
    shutil.rmtree(out, ignore_errors=True)
    subprocess.check_call([exe, script_path, out], cwd=WORK, stdout=subprocess.DEVNULL)
    return out


def collect(out, sample=None, seed=12345):
    st = json.load(open(os.path.join(out, "stats.json")))
    W, H = st["width"], st["height"]
    rgb = np.fromfile(os.path.join(out, "rgb.f32"), np.float32).reshape(H, W, 3)
    packed = np.fromfile(os.path.join(out, "packed.u32"), np.uint32).reshape(H, W)
    dist = np.fromfile(os.path.join(out, "dist.f32"), np.float32).reshape(H, W)
    assert st["packed_mismatch"] in (0, -1), st
    g = {"stats": np.array([st["primary"], st["shadow"], st["reflection"], st["refraction"]], np.int64),
         "size": np.array([W, H], np.int32), "bounces": np.int32(st["bounces"]),
         "channel_sums": rgb.astype(np.float64).sum(axis=(0, 1))}
    if sample:
        rng = np.random.RandomState(seed)
        xs = rng.randint(0, W, sample).astype(np.int32)
        ys = rng.randint(0, H, sample).astype(np.int32)
        g.update(xs=xs, ys=ys, rgb=rgb[ys, xs], packed=packed[ys, xs], dist=dist[ys, xs])
    else:
        g.update(rgb=rgb, packed=packed, dist=dist)
    k = 0
    while os.path.exists(os.path.join(out, f"texprobe{k}.f32")):
        g[f"texprobe{k}"] = np.fromfile(os.path.join(out, f"texprobe{k}.f32"), np.float32).reshape(-1, 9)
        k += 1
    return g


def save_scene_gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", compresslevel=9, mtime=0) as g:
        g.write(f.read())


def script_with(name, extra_lines):
    base = open(os.path.join(HERE, "scenes", name + ".txt")).read()
    p = os.path.join(WORK, name + "_gen.txt")
    with open(p, "w") as f:
        f.write(base + "\n" + "\n".join(extra_lines) + "\n")
    return p


def main():
    if not os.path.isdir(REF):
        sys.exit("make_goldens.py needs /root/reference (build container only)")
    stage()
    os.makedirs(OUT, exist_ok=True)

    def golden(name, script, variant, sample=None, extra=(), keep_scene=True, tag=None):
        out = run_ref(variant, script_with(script, list(extra)), name)
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        if keep_scene:
            save_scene_gz(os.path.join(out, "scene.rtxs"), os.path.join(d, "scene.rtxs.gz"))
        g = collect(out, sample)
        np.savez_compressed(os.path.join(d, f"golden{'_' + tag if tag else ''}.npz"), **g)
        print(name, tag or "", variant, "stats", g["stats"].tolist())
        return out

    tp = ["texprobe ./Data/Floor.png 2048 12345", "texprobe ./Data/LEGOSHLD.tga 2048 777"]
    mat_extra = ["size 320 180", "matset mesh:0:0 texture ./Data/LEGOSHLD.tga"] + tp

    def knob_goldens():
        # the remaining Config.h values, each as its own build of the reference on the `materials` scene (scene file shared with the
        # other materials goldens; tests/util.py applies the config override): BVH_TRAVERSE_TREE_NAIVE (Config.h:27),
        # TEXTURE_SAMPLE_MODE_NEAREST (:38), MAX_ANISOTROPY 2 (:55), NUMBER_OF_BOUNCES 1 and 5 (:12)
        for tag, variant, more in (("naive", "s0_m1_b3_t2_naive", []), ("nearest", "s0_m1_b3_t0", []), ("aniso2", "s0_m1_b3_t2_a2", tp),
                                   ("b1", "s0_m1_b1_t2", ["bounces 1"]), ("b5", "s0_m1_b5_t2", ["bounces 5"])):
            golden("materials", "materials", variant, extra=mat_extra[:2] + more, tag=tag, keep_scene=False)
    if "--knobs-only" in sys.argv:
        knob_goldens()
        return
    # BVH_VISUALIZE_HEATMAP builds (Config.h:23, Raytracer.cpp:97-102): same scenes, pixels = BLAS traversal steps
    golden("monkey_small", "monkey_small", "s0_m1_b3_t2_heat", keep_scene=False, tag="heat")
    golden("materials", "materials", "s0_m1_b3_t2_heat", extra=mat_extra[:2], keep_scene=False, tag="heat")
    golden("camera_keys", "camera_keys", "s0_m1_b3_t2")      # Camera::update with keys held, through the harness's `keys` command
    if "--heat-only" in sys.argv or "--extras-only" in sys.argv:
        return

    golden("cube", "cube", "s0_m1_b0_t2")
    golden("monkey", "monkey", "s0_m1_b3_t2", sample=8192)
    golden("monkey_small", "monkey_small", "s0_m1_b3_t2")

    o_an = golden("materials", "materials", "s0_m1_b3_t2", extra=mat_extra, tag="aniso")
    o_tr = golden("materials", "materials", "s0_m0_b3_t2", extra=mat_extra, tag="trilinear", keep_scene=False)
    o_ew = golden("materials", "materials", "s0_m2_b3_t2", extra=mat_extra, tag="ewa", keep_scene=False)
    o_bi = golden("materials", "materials", "s0_m1_b3_t1", extra=mat_extra, tag="bilinear", keep_scene=False)
    # the three mip variants must describe the same scene up to config.mip_filter; bilinear mode drops the mip chains
    a = sio.load_scene(os.path.join(o_an, "scene.rtxs"))
    for o in (o_tr, o_ew):
        b = sio.load_scene(os.path.join(o, "scene.rtxs"))
        assert all(np.array_equal(x.texels, y.texels) for x, y in zip(a.textures, b.textures))
        assert np.array_equal(a.instances, b.instances) and np.array_equal(a.materials, b.materials)
    b = sio.load_scene(os.path.join(o_bi, "scene.rtxs"))
    for x, y in zip(a.textures, b.textures):
        n0 = int(x.desc["width"][0]) * int(x.desc["height"][0])
        assert np.array_equal(x.texels[:n0], y.texels[:n0]) and int(y.desc["mipmapped"][0]) == 0

    knob_goldens()
    golden("dynamic", "dynamic", "s1_m2_b3_t2", extra=["size 320 180"])
    golden("tori16", "tori16", "s1_m2_b3_t2", extra=["size 320 180"])
    golden("tori16_f1", "tori16", "s1_m2_b3_t2", extra=["size 320 180", "frames 1 0.0166666667"])

    total = 0
    for root, _, files in os.walk(OUT):
        for f in files:
            total += os.path.getsize(os.path.join(root, f))
    print("tests/golden total bytes:", total)


if __name__ == "__main__":
    main()
