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
from pyrtx import host, obj_export, scene_io as sio  # noqa: E402
import make_goldens as mg  # noqa: E402
import orc  # noqa: E402


def reachable_nodes(nodes):
    """Mask of the nodes a traversal can reach (the reference leaves index 1 and abandoned child pairs unwritten)."""
    cnt = nodes["count"]; lf = nodes["left_or_first"]
    seen = np.zeros(len(nodes), bool); stack = [0]
    while stack:
        k = stack.pop(); seen[k] = True
        if (int(cnt[k]) & 0x3fffffff) == 0:
            stack += [int(lf[k]), int(lf[k]) + 1]
    return seen


def export(detail, seed, d):
    pos, nrm, uv, mid, nmat = host.atrium_mesh(seed, detail)
    obj_export.write_obj(d, "atrium", pos, nrm, uv, mid, host.atrium_materials(nmat), host.procedural_texture_images(seed))
    return len(pos)


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

    def per_ray(w):
        return {"nodes_per_closest_ray": round((w["blas_nodes_closest"]) / w["closest_rays"], 2), "tris_per_closest_ray": round(w["tri_tests_closest"] / w["closest_rays"], 2),
                "nodes_per_shadow_ray": round(w["blas_nodes_any"] / max(w["any_rays"], 1), 2), "tris_per_shadow_ray": round(w["tri_tests_any"] / max(w["any_rays"], 1), 2)}

    def own_tree(accel):
        sc_own = host.atrium_scene(W, H, 3, detail=detail, seed=seed, accel=accel)
        own = orc.OracleScene(sc_own).render(threads=8)
        d_rgb = np.abs(own["rgb"].astype(np.float64) - ref_rgb.astype(np.float64))
        return {"bvh_nodes": int(len(sc_own.blas[0].nodes)), "references": int(len(sc_own.blas[0].tri_hot)), "stats": own["stats"], "traversal": per_ray(own["work"]),
                "pixels_differing_vs_reference": int((d_rgb.max(axis=-1) > 0).sum()), "pixels_differing_gt_1e-4": int((d_rgb.max(axis=-1) > 1e-4).sum()),
                "max_abs": float(np.nanmax(d_rgb)), "packed_mismatch": int((own["packed"] != ref_packed).sum()),
                "note": "OBJ text round-trips vertex data through tinyobjloader's float parser, so geometry is equal only to ~1 ulp between the two scenes"}
    res = {
        "workload": f"atrium detail {detail}: {n} triangles, {W}x{H}, depth 3, 3 lights",
        "reference": {"render_tile_ms_1thread_lane1": st["render_tile_ms_1thread"], "stats": [st["primary"], st["shadow"], st["reflection"], st["refraction"]],
                      "sbvh_nodes": int(len(sc_ref.blas[0].nodes)), "sbvh_refs": int(len(sc_ref.blas[0].tri_hot)), "wall_s_incl_sbvh_build": round(t_ref, 1)},
        "oracle_on_reference_sbvh": {"bit_exact_rgb_and_packed": exact, "stats_equal": stats_ok, "traversal": per_ray(o["work"])},
        "oracle_on_restated_sbvh": own_tree("sbvh"),      # bench default: librtx_host's SBVH on the procedural mesh itself
        "oracle_on_own_bvh": own_tree("binned"),          # this repo's binned-SAH builder
    }
    # fixture for tests/test_sbvh.py: digest of the SBVH the reference built for this mesh (its own .bvh cache file)
    import hashlib
    rb = host.load_bvh_cache(os.path.join(d, "atrium.obj.bvh"))
    digest = {"detail": detail, "seed": seed, "source_triangles": n, "nodes": int(len(rb.nodes)), "references": int(len(rb.tri_hot)),
              "reachable_nodes_sha256": hashlib.sha256(rb.nodes[reachable_nodes(rb.nodes)].tobytes()).hexdigest(),
              "tri_hot_sha256": hashlib.sha256(rb.tri_hot.tobytes()).hexdigest(),
              "tri_cold_sha256": hashlib.sha256(rb.tri_cold.tobytes()).hexdigest(),
              "made_by": "oracle/ref_harness/atrium_crosscheck.py from the .bvh cache written by the real reference (MESH_ACCELERATOR_SBVH)"}
    json.dump(digest, open(os.path.join(REPO, "tests", "golden", f"atrium_d{detail}_ref_sbvh.json"), "w"), indent=1)
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "profiles", "r01_atrium_crosscheck.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
