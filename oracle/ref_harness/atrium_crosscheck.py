#!/usr/bin/env python3
"""cfg3-scale cross-check with the REAL reference (build container only; test infrastructure).

The cfg3 workload (Sponza-class "atrium" stand-in) is built on the GPU box with this repo's own BVH
builder, so its scene file cannot be a committed fixture.  This script closes the loop here:
  1. exports the procedural mesh as OBJ / MTL / TGA into the scratch Data directory,
  2. lets the reference load it, build its OWN SBVH and render 1920x1080 at depth 3 (refdump),
  3. checks the oracle BIT-EXACT against that frame on the reference-built SBVH (oracle pinned at scale),
  4. renders the same camera with this repo's binned-SAH BVH on the oracle and reports how many pixels
     differ between the two acceleration structures (closest hits are topology-independent except for
     exact-t ties / ulp-level culling), plus traversal statistics of both trees.
Writes profiles/r01_atrium_crosscheck.json.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(REPO, "cpu-raytracer_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, HERE)
from pyrtx import host, scene_io as sio  # noqa: E402
import make_goldens as mg  # noqa: E402
import orc  # noqa: E402


def write_tga(path, img_u8):
    h, w, _ = img_u8.shape
    hdr = bytearray(18); hdr[2] = 2; hdr[12] = w & 255; hdr[13] = w >> 8; hdr[14] = h & 255; hdr[15] = h >> 8; hdr[16] = 24; hdr[17] = 0x20   # top-left origin
    with open(path, "wb") as f:
        f.write(bytes(hdr)); f.write(img_u8[:, :, ::-1].tobytes())


def export(detail, seed, d):
    os.makedirs(d, exist_ok=True)
    pos, nrm, uv, mid, nmat = host.atrium_mesh(seed, detail)
    mats = host.atrium_materials(nmat)
    for i, img in enumerate(host.procedural_texture_images(seed)):
        write_tga(os.path.join(d, f"tex{i}.tga"), img)
    with open(os.path.join(d, "atrium.mtl"), "w") as f:
        for i, m in enumerate(mats):
            f.write(f"newmtl m{i}\nKd {m['diffuse'][0]:.9g} {m['diffuse'][1]:.9g} {m['diffuse'][2]:.9g}\n"
                    f"Ks {m['reflection'][0]:.9g} {m['reflection'][1]:.9g} {m['reflection'][2]:.9g}\n"
                    f"Kt {m['transmittance'][0]:.9g} {m['transmittance'][1]:.9g} {m['transmittance'][2]:.9g}\nNi {m['index_of_refraction']:.9g}\n")
            if m["texture_id"] >= 0:
                f.write(f"map_Kd tex{int(m['texture_id'])}.tga\n")
            f.write("\n")
    n = len(pos)
    with open(os.path.join(d, "atrium.obj"), "w") as f:
        f.write("mtllib atrium.mtl\n")
        p = pos.reshape(-1, 3); nn = nrm.reshape(-1, 3); t = uv.reshape(-1, 2).copy()
        t[:, 1] = 1.0 - t[:, 1]                                   # the loader flips v back (OBJLoader.cpp:139-141)
        f.write("".join(f"v {a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in p))
        f.write("".join(f"vt {a:.9g} {b:.9g}\n" for a, b in t))
        f.write("".join(f"vn {a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in nn))
        order = np.argsort(mid, kind="stable")
        cur = -1
        out = []
        for k in order:
            if mid[k] != cur:
                cur = int(mid[k]); out.append(f"usemtl m{cur}\n")
            a = 3 * k + 1
            out.append(f"f {a}/{a}/{a} {a+1}/{a+1}/{a+1} {a+2}/{a+2}/{a+2}\n")
        f.write("".join(out))
    return n


def main():
    detail = int(os.environ.get("ATRIUM_DETAIL", "1")); seed = 0x5EED0003
    W, H = (int(x) for x in os.environ.get("ATRIUM_SIZE", "1920x1080").split("x"))
    mg.stage()
    d = os.path.join(mg.WORK, "Data", "atrium")
    for f in ("atrium.obj.bvh",):
        try: os.remove(os.path.join(d, f))
        except OSError: pass
    n = export(detail, seed, d)
    cam, pt, sp, dr = host.ATRIUM_CAMERA, host.ATRIUM_POINT, host.ATRIUM_SPOT, host.ATRIUM_DIR
    script = os.path.join(mg.WORK, "atrium_gen.txt")
    with open(script, "w") as f:
        f.write(f"size {W} {H}\nbounces 3\nmesh ./Data/atrium/atrium.obj 0 0 0\n")
        f.write("point %g %g %g %g %g %g\n" % (*pt[0], *pt[1]))
        f.write("spot %g %g %g %g %g %g %.9g %.9g %.9g %g %g\n" % (*sp[0], *sp[1], *sp[2], sp[3], sp[4]))
        f.write("dir %g %g %g %.9g %.9g %.9g\n" % (*dr[0], *dr[1]))
        f.write("camera_axis_angle %g %g %g %g %g %g %.9g\n" % (*cam[0], *cam[1][0], cam[1][1]))
    t0 = time.time()
    out = mg.run_ref("s0_m1_b3_t2", script, "atrium")
    t_ref = time.time() - t0
    st = json.load(open(os.path.join(out, "stats.json")))
    ref_rgb = np.fromfile(os.path.join(out, "rgb.f32"), np.float32).reshape(H, W, 3)
    ref_packed = np.fromfile(os.path.join(out, "packed.u32"), np.uint32).reshape(H, W)
    sc_ref = sio.load_scene(os.path.join(out, "scene.rtxs"))
    o = orc.OracleScene(sc_ref).render(threads=8)
    exact = bool(np.array_equal(o["rgb"].view(np.uint32), ref_rgb.view(np.uint32)) and np.array_equal(o["packed"], ref_packed))
    stats_ok = [o["stats"][k] for k in ("primary", "shadow", "reflection", "refraction")] == [st["primary"], st["shadow"], st["reflection"], st["refraction"]]

    sc_own = host.atrium_scene(W, H, 3, detail=detail, seed=seed)
    own = orc.OracleScene(sc_own).render(threads=8)
    d_rgb = np.abs(own["rgb"].astype(np.float64) - ref_rgb.astype(np.float64))
    def per_ray(w):
        return {"nodes_per_closest_ray": round((w["blas_nodes_closest"]) / w["closest_rays"], 2), "tris_per_closest_ray": round(w["tri_tests_closest"] / w["closest_rays"], 2),
                "nodes_per_shadow_ray": round(w["blas_nodes_any"] / max(w["any_rays"], 1), 2), "tris_per_shadow_ray": round(w["tri_tests_any"] / max(w["any_rays"], 1), 2)}
    res = {
        "workload": f"atrium detail {detail}: {n} triangles, {W}x{H}, depth 3, 3 lights",
        "reference": {"render_tile_ms_1thread_lane1": st["render_tile_ms_1thread"], "stats": [st["primary"], st["shadow"], st["reflection"], st["refraction"]],
                      "sbvh_nodes": int(len(sc_ref.blas[0].nodes)), "sbvh_refs": int(len(sc_ref.blas[0].tri_hot)), "wall_s_incl_sbvh_build": round(t_ref, 1)},
        "oracle_on_reference_sbvh": {"bit_exact_rgb_and_packed": exact, "stats_equal": stats_ok, "traversal": per_ray(o["work"])},
        "oracle_on_own_bvh": {"bvh_nodes": int(len(sc_own.blas[0].nodes)), "stats": own["stats"], "traversal": per_ray(own["work"]),
                              "pixels_differing_vs_reference": int((d_rgb.max(axis=-1) > 0).sum()), "pixels_differing_gt_1e-4": int((d_rgb.max(axis=-1) > 1e-4).sum()),
                              "max_abs": float(np.nanmax(d_rgb)), "packed_mismatch": int((own["packed"] != ref_packed).sum()),
                              "note": "OBJ text round-trips vertex data through tinyobjloader's float parser, so geometry is equal only to ~1 ulp between the two scenes"},
    }
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "profiles", "r01_atrium_crosscheck.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
