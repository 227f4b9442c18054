#!/usr/bin/env python3
"""Per-function unit vectors from the REAL reference: TEST INFRASTRUCTURE, build container only.

Runs oracle/_ref/refdump_s0_m1_b3_t2 on the `materials` golden scene (meshes, dielectric spheres, textured plane) with the harness
command `unitprobe N seed`, which calls one reference function per record at seeded inputs:
  rays   Scene::trace_primitives -> every RayHit field (RayHit.h:5-21: hit, distance, point, normal, material_id, u, v, ds/dt_dx/dy,
         dO_dx/dy, dN_dx/dy) through TopLevelBVH::trace, Mesh::trace, BottomLevelBVH::trace + triangle_trace, Sphere::trace, Plane::trace;
         Scene::intersect_primitives for two maximum distances
  aabb   AABB::intersect incl. flat boxes, origins on slab planes and zero direction components (the 0 * inf cases)
  light  PointLight / SpotLight / DirectionalLight::calc_lighting
  plot   Window::plot
The scene the rays were traced against is the committed golden scene tests/golden/materials/scene.rtxs.gz (checked here byte for byte).
Output: tests/golden/unit/unit_vectors.npz."""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402

N, SEED = 2048, 20261004


def main():
    if not os.path.isdir(mg.REF):
        sys.exit("needs /root/reference (build container only)")
    mg.stage()
    extra = ["size 320 180", "matset mesh:0:0 texture ./Data/LEGOSHLD.tga", f"unitprobe {N} {SEED}"]
    out = mg.run_ref("s0_m1_b3_t2", mg.script_with("materials", extra), "unit")
    committed = gzip.open(os.path.join(mg.OUT, "materials", "scene.rtxs.gz"), "rb").read()
    assert committed == open(os.path.join(out, "scene.rtxs"), "rb").read(), "the probed scene is not the committed materials scene"
    d = os.path.join(mg.OUT, "unit")
    os.makedirs(d, exist_ok=True)
    rd = lambda name, w: np.fromfile(os.path.join(out, name), np.float32).reshape(-1, w)
    rays, aabb, light, plot = rd("unit_rays.f32", 47), rd("unit_aabb.f32", 14), rd("unit_light.f32", 19), rd("unit_plot.f32", 4)
    np.savez_compressed(os.path.join(d, "unit_vectors.npz"), rays=rays, aabb=aabb, light=light, plot_rgb=plot[:, :3].copy(), plot_packed=plot[:, 3].copy().view(np.uint32))
    hit = rays[:, 18] > 0
    print("rays", len(rays), "hits", int(hit.sum()), "materials hit", np.unique(rays[hit, 26].astype(int)).tolist(), "occluded@3", int(rays[:, 45].sum()), "occluded@inf", int(rays[:, 46].sum()))
    print("aabb pass", int(aabb[:, 13].sum()), "of", len(aabb), "nan inv", int(np.isinf(aabb[:, 9:12]).any(axis=1).sum()))
    print("light nonzero point/spot/dir", int((light[:, 10:13] != 0).any(axis=1).sum()), int((light[:, 13:16] != 0).any(axis=1).sum()), int((light[:, 16:19] != 0).any(axis=1).sum()))
    print("bytes", os.path.getsize(os.path.join(d, "unit_vectors.npz")))


if __name__ == "__main__":
    main()
