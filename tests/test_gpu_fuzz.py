"""Differential fuzz: random small scenes (meshes from the data files and random triangle soups, random instance transforms,
spheres, planes, all three light kinds, dielectric / mirror / textured materials, axis-aligned and general cameras) rendered
by the HIP path and by the oracle.  The oracle is pinned to the reference on the goldens; this spreads the GPU == oracle check
over inputs no golden covers (exact zeros in ray directions -> the NaN slab-test path, degenerate triangles, deep instancing)."""
import os

import numpy as np
import pytest

import util
from pyrtx import assemble, host, scene_io as sio

pytestmark = pytest.mark.gpu
DATA = os.path.join(util.GOLDEN, "meshes")
MESHES = ["Cube", "Diamond", "icosphere", "Concave", "Torus", "Monkey", "Rock"]


def random_scene(seed):
    rng = np.random.default_rng(seed)
    f = lambda lo, hi: float(rng.uniform(lo, hi))
    lines = [f"size {int(rng.integers(40, 130))} {int(rng.integers(33, 90))}", f"bounces {int(rng.integers(0, 4))}"]
    n_mesh = int(rng.integers(1, 6))
    for _ in range(n_mesh):
        m = MESHES[int(rng.integers(0, len(MESHES)))]
        if rng.random() < 0.3:
            lines.append(f"mesh ./Data/{m}.obj {f(-4, 4):.4f} {f(-1, 3):.4f} {f(2, 10):.4f}")                # identity rotation: exact zeros survive
        else:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            lines.append(f"mesh_axis_angle ./Data/{m}.obj {f(-4, 4):.4f} {f(-1, 3):.4f} {f(2, 10):.4f} {ax[0]:.6f} {ax[1]:.6f} {ax[2]:.6f} {f(-3, 3):.4f}")
    for i in range(n_mesh):
        if rng.random() < 0.5:
            lines.append(f"matset mesh:{i}:0 reflection {f(0, 0.8):.3f} {f(0, 0.8):.3f} {f(0, 0.8):.3f}")
        if rng.random() < 0.3:
            lines.append(f"matset mesh:{i}:0 transmittance {f(0, 1):.3f} {f(0, 1):.3f} {f(0, 1):.3f}")
            lines.append(f"matset mesh:{i}:0 ior {f(1.0, 2.4):.3f}")
        if rng.random() < 0.4:
            lines.append(f"matset mesh:{i}:0 texture ./Data/{'Floor.png' if rng.random() < 0.5 else 'LEGOSHLD.tga'}")
    for i in range(int(rng.integers(0, 3))):
        lines.append(f"sphere {f(-3, 3):.3f} {f(0, 2):.3f} {f(2, 7):.3f} {f(0.3, 1.2):.3f}")
        lines.append(f"matset sphere:{i} transmittance {f(0, 1):.3f} {f(0, 1):.3f} {f(0, 1):.3f}")
        lines.append(f"matset sphere:{i} ior {f(1.0, 1.8):.3f}")
        lines.append(f"matset sphere:{i} reflection {f(0, 0.5):.3f} {f(0, 0.5):.3f} {f(0, 0.5):.3f}")
    if rng.random() < 0.6:
        lines.append("plane 0 -1.5 0" if rng.random() < 0.5 else f"plane_axis_angle 0 {f(-2.5, -1):.3f} 0 0 1 0 {f(0, 3):.3f}")
        lines.append("matset plane:0 texture ./Data/Floor.png")
        lines.append(f"matset plane:0 reflection {f(0, 0.4):.3f} {f(0, 0.4):.3f} {f(0, 0.4):.3f}")
    if rng.random() < 0.7:
        lines.append(f"point {f(2, 30):.2f} {f(2, 30):.2f} {f(2, 30):.2f} {f(-5, 5):.3f} {f(2, 8):.3f} {f(-2, 8):.3f}")
    if rng.random() < 0.5:
        lines.append(f"spot {f(5, 40):.2f} {f(5, 40):.2f} {f(5, 40):.2f} {f(-5, 5):.3f} {f(3, 8):.3f} {f(-2, 4):.3f} {f(-1, 1):.3f} -1 {f(-0.5, 1):.3f} {f(20, 50):.1f} {f(55, 90):.1f}")
    if rng.random() < 0.7:
        d = [(0, -1, 0), (1, -1, 0), (0, -1, 1)][int(rng.integers(0, 3))] if rng.random() < 0.5 else (f(-1, 1), -1, f(-1, 1))   # axis-aligned: zero direction components
        lines.append(f"dir {f(0.2, 1):.3f} {f(0.2, 1):.3f} {f(0.2, 1):.3f} {d[0]} {d[1]} {d[2]}")
    if rng.random() < 0.3:
        lines.append("camera 0 1 -4 0 0 0 1")                                                                # identity: the centre column has direction x == 0
    else:
        lines.append(f"camera_axis_angle {f(-2, 2):.3f} {f(0, 3):.3f} {f(-5, -2):.3f} {f(-0.3, 0.3):.3f} 1 0 {f(-0.5, 0.5):.3f}")
    mip = int(rng.integers(0, 3)); tex_mode = 2 if rng.random() < 0.8 else int(rng.integers(0, 2))
    return "\n".join(lines) + "\n", mip, tex_mode


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTX_FUZZ_SEEDS", "24"))))       # RTX_FUZZ_SEEDS=300 for a long soak
def test_random_scene_gpu_equals_oracle(seed):
    import orc
    from pyrtx import api
    text, mip, tex_mode = random_scene(1000 + seed)
    sc = assemble.scene_from_script(text, DATA, accel=["sbvh", "bvh", "binned"][seed % 3], mip_filter=mip, texture_mode=tex_mode)
    ref = orc.OracleScene(sc).render(threads=16)
    r = api.Renderer(sc)
    out = r.render()
    assert out["stats"] == ref["stats"], text
    same_nan = np.isnan(out["rgb"]) == np.isnan(ref["rgb"])
    assert same_nan.all(), text
    d = np.abs(out["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64)); d[np.isnan(d)] = 0
    assert d.max() < 1e-4, (text, float(d.max()))
    assert util.bit_exact(out["rgb"], ref["rgb"]), (text, int((d.max(axis=-1) > 0).sum()))
    assert np.array_equal(out["packed"], ref["packed"]), text
    simple = api.Renderer(sc).render(simple_trace=True)                   # plain reference-order kernel == production kernel
    assert np.array_equal(simple["rgb"].view(np.uint32), out["rgb"].view(np.uint32)), text
    # every other launch shape / kernel choice: the one-stream shape bench.py times (merged shadow-ray launch), per-lane shadow
    # rays, packet closest-hit rays
    for kw in ({"serial": True}, {"serial": True, "lane_trace": True}, {"packet_closest": True}, {"serial": True, "packet_closest": True}):
        alt = r.render(**kw)
        assert np.array_equal(alt["rgb"].view(np.uint32), out["rgb"].view(np.uint32)) and alt["stats"] == out["stats"], (kw, text)
