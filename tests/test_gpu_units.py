"""Unit-level parity of single reference functions evaluated on the device (C ABI debug hooks) and
edge-case configurations of the whole path."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from pyrtx import api as a
    a.load_library()
    return a


@pytest.mark.parametrize("name", ["materials_aniso", "materials_trilinear", "materials_ewa", "materials_bilinear", "materials_aniso2"])
def test_texture_sample_matches_reference_probes(api, name):
    """Texture::sample (Texture.h:33-49) at 2 x 2048 seeded (s,t,derivative) inputs recorded from the REAL reference.
    Bilinear / trilinear / anisotropic / EWA.  log2f (mip LOD) is evaluated in fp64 on the device, so a sample whose
    LOD lands within 1 ulp of a level boundary could pick another level; none may exceed the tolerance here."""
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    # probe k was recorded on texture file k of the script: Floor.png then LEGOSHLD.tga; scene textures are in first-use order
    sizes = [(int(t.desc["width"][0]), int(t.desc["height"][0])) for t in sc.textures]
    for k, wh in enumerate([(32, 32), (256, 256)]):
        probe = g[f"texprobe{k}"]
        out = r.debug_texture_sample(sizes.index(wh), probe[:, :6])
        d = np.abs(out.astype(np.float64) - probe[:, 6:9])
        assert np.nanmax(d) < 1e-6, (name, k, float(np.nanmax(d)))
        assert (d.max(axis=1) > 0).mean() < 0.02          # almost all samples are bit-exact


def test_sky_sample_matches_oracle_bit_exact(api):
    import ctypes as C
    import orc
    sc, _ = util.load_golden("cube")
    r = api.Renderer(sc)
    rng = np.random.RandomState(11)
    d = rng.normal(size=(100000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    dev = r.debug_sky_sample(d)
    o = orc.OracleScene(sc)
    ref = np.zeros_like(d)
    lib = orc.lib()
    for i in range(0, len(d), 1):
        lib.orc_sky_sample(o._sky.ctypes.data, o.struct.sky_size, d[i].ctypes.data, ref[i].ctypes.data)
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("override", [{"traversal_strategy": 0}, {"texture_mode": 0}, {"bounces": 0}, {"bounces": 1}, {"bounces": 5},
                                      {"stack_size": 40}, {"max_anisotropy": 2.0}])
def test_config_knobs_vs_oracle(api, override):
    """Every Config.h knob (Config.h:1-55) that changes the result: GPU vs oracle on the dielectric/texture scene."""
    import orc
    sc, _ = util.load_golden("materials_aniso")
    for k, v in override.items():
        sc.config[k] = v
    if override.get("texture_mode", 2) != 2:
        for t in sc.textures:
            t.desc["mipmapped"] = 0; t.desc["mip_levels"] = 1
    out = api.Renderer(sc).render()
    ref = orc.OracleScene(sc).render(threads=8)
    assert out["stats"] == ref["stats"], override
    d = np.abs(out["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
    assert np.nanmax(d) < 1e-4 and np.array_equal(out["packed"], ref["packed"]), override


def test_multi_batch_frame_equals_single_batch(api, monkeypatch):
    """Deep recursion is rendered in tile batches bounded by the queue budget (plan_batch)."""
    sc, _ = util.load_golden("materials_aniso")
    full = api.Renderer(sc).render()
    monkeypatch.setenv("RTX_SLOT_BUDGET", str(1024 * 15 * 7))      # 7 tiles per batch at depth 3
    part = api.Renderer(sc).render()
    assert np.array_equal(full["rgb"].view(np.uint32), part["rgb"].view(np.uint32)) and full["stats"] == part["stats"]


def test_nan_from_beer_law_is_reproduced(api):
    """Kt == 1 with a refracted ray that misses gives exp(0 * inf) = NaN in the reference (Raytracer.cpp:370-376)."""
    import orc
    sc, _ = util.load_golden("materials_aniso")
    pm = int(sc.planes["material_id"][0])          # rays entering the (open) plane refract below it and hit nothing
    sc.materials["transmittance"][pm] = (1.0, 1.0, 1.0); sc.materials["index_of_refraction"][pm] = 1.2
    out = api.Renderer(sc).render()
    ref = orc.OracleScene(sc).render(threads=8)
    assert np.isnan(ref["rgb"]).any()
    assert np.array_equal(np.isnan(out["rgb"]), np.isnan(ref["rgb"]))
    m = ~np.isnan(ref["rgb"])
    assert np.max(np.abs(out["rgb"][m] - ref["rgb"][m])) < 1e-4
    assert np.array_equal(out["packed"], ref["packed"])


def test_stack_overflow_is_reported_not_silent(api):
    """A degenerate chain-shaped BVH deeper than BVH_TRAVERSAL_STACK_SIZE makes the reference smash its (unchecked) stack; here the tree is
    refused before anything is launched (RTX_ERR_LIMIT from rtx_render_tiles: an inner node at depth 79 needs 81 entries)."""
    from pyrtx import scene_io as sio
    sc, _ = util.load_golden("cube")
    depth = 80
    nodes = np.zeros(2 + 2 * depth, sio.BVH_NODE)     # root 0, (unused 1), pairs (2d+2, 2d+3)
    hot, cold = sc.blas[0].tri_hot[:1].copy(), sc.blas[0].tri_cold[:1].copy()
    for d in range(depth):
        k = 0 if d == 0 else 2 * d              # the LEFT child continues the chain: its right sibling (a leaf) waits on the stack
        nodes[k]["aabb_min"] = -2; nodes[k]["aabb_max"] = 2
        nodes[k]["left_or_first"] = 2 * d + 2; nodes[k]["count"] = 1 << 30
        leaf = 2 * d + 3
        nodes[leaf]["aabb_min"] = -2; nodes[leaf]["aabb_max"] = 2; nodes[leaf]["left_or_first"] = 0; nodes[leaf]["count"] = 1
    last = 2 * depth
    nodes[last]["aabb_min"] = -2; nodes[last]["aabb_max"] = 2; nodes[last]["left_or_first"] = 0; nodes[last]["count"] = 1
    sc.blas[0] = sio.Blas(nodes, hot, cold, sc.blas[0].material_offset, 1)
    sc.config["traversal_strategy"] = 0          # naive: always left first
    r = api.Renderer(sc)
    with pytest.raises(api.RtxError) as e:
        r.render_async()
    assert e.value.code == 4 and "stack" in str(e.value)


def test_dangling_material_and_texture_ids_are_refused_before_launch(api):
    """A material id past the table or a texture id never uploaded would make a kernel read out of bounds; the C ABI must answer
    with RTX_ERR_STATE from rtx_render_tiles instead (the reference would index MaterialBuffer::materials blindly)."""
    import copy
    base, _ = util.load_golden("materials_aniso")
    def attempt(mutate):
        sc = copy.deepcopy(base)
        mutate(sc)
        r = api.Renderer(sc)
        with pytest.raises(api.RtxError) as e:
            r.render_async()
        assert e.value.code == 5, e.value
    def bad_sphere(sc): sc.spheres["material_id"][0] = len(sc.materials) + 3
    def bad_plane(sc): sc.planes["material_id"][0] = -2
    def bad_triangle(sc): sc.blas[0].tri_cold["material_id"][0] = 200
    def bad_texture(sc): sc.materials["texture_id"][1] = 77
    def short_table(sc): sc.materials = sc.materials[:2].copy()
    for m in (bad_sphere, bad_plane, bad_triangle, bad_texture, short_table):
        attempt(m)
    ok = api.Renderer(copy.deepcopy(base)).render()                       # the untouched scene still renders
    assert ok["stats"]["primary"] == base.width * base.height


def test_cyclic_node_arrays_are_refused(api):
    """An inner node whose child pair does not lie behind it could send a traversal round in circles for ever (a hung GPU);
    rtx_upload_blas / rtx_set_frame refuse such arrays — every builder of the reference allocates children after their parent."""
    import copy
    base, _ = util.load_golden("monkey_small")
    sc = copy.deepcopy(base)
    inner = [k for k in range(len(sc.blas[0].nodes)) if k != 1 and (int(sc.blas[0].nodes[k]["count"]) & 0x3fffffff) == 0]
    sc.blas[0].nodes["left_or_first"][inner[5]] = 0                        # points back at the root pair
    with pytest.raises(api.RtxError) as e:
        api.Renderer(sc)
    assert e.value.code == 1
    sc = copy.deepcopy(base)
    if len(sc.tlas_nodes) > 2 and (int(sc.tlas_nodes[0]["count"]) & 0x3fffffff) == 0:
        sc.tlas_nodes["left_or_first"][0] = 0
        with pytest.raises(api.RtxError):
            api.Renderer(sc)


@pytest.mark.parametrize("name,k", [("cube", 1), ("cube", 2), ("cube", 12), ("monkey_small", 40), ("monkey_small", 300)])
def test_single_leaf_mesh(api, name, k):
    """A mesh whose BVH is one leaf (the root IS the leaf, node_count 1): the packet walkers enter their leaf path straight from the
    instance entry, the 4-wide records are empty; a leaf of 300 triangles exceeds their packed entries, so that mesh keeps the binary
    walk (the closest-hit per-lane walk's ordered records: from 16 triangles).  Frame == oracle in every launch shape."""
    import copy
    import orc
    sc, _ = util.load_golden(name)
    sc = copy.deepcopy(sc)
    b = sc.blas[0]
    hot = b.tri_hot[:k].copy(); cold = b.tri_cold[:k].copy()
    pts = np.concatenate([hot["position_0"], hot["position_0"] + hot["position_edge_1"], hot["position_0"] + hot["position_edge_2"]]).astype(np.float32)
    nodes = np.zeros(1, util.sio.BVH_NODE)
    nodes["aabb_min"][0] = pts.min(axis=0); nodes["aabb_max"][0] = pts.max(axis=0)
    nodes["left_or_first"][0] = 0; nodes["count"][0] = k
    b.nodes = nodes; b.tri_hot = hot; b.tri_cold = cold
    ref = orc.OracleScene(sc).render(threads=4)
    r = api.Renderer(sc)
    assert r.debug_blas_wide(0) == (0 if k < 256 else -1)
    assert r.debug_blas_wide_closest(0) == (0 if k < 16 else -1)      # stack and work-list entries of the closest-hit walk pack leaves of up to 15 triangles
    for mode in ({}, {"serial": True}, {"lane_trace": True}, {"packet_closest": True}):
        out = r.render(**mode)
        assert out["stats"] == ref["stats"], mode
        assert util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"]), mode


def test_group_attach_and_debug_entry_points_refuse_bad_arguments(api):
    """Status codes instead of faults: rtx_group_attach with a foreign resolution / the base itself / no group, rtx_debug_blas_wide with an
    unknown mesh id."""
    import ctypes as C
    sc, _ = util.load_golden("cube")
    sc2, _ = util.load_golden("monkey_small")
    r, r_same, r_other = api.Renderer(sc), api.Renderer(sc), api.Renderer(sc2)
    r.group_create(0, 1, api.Renderer.group_unique_id())
    out = C.c_void_p()
    assert r.lib.rtx_group_attach(r.ctx, r.group, C.byref(out)) == 1              # the base context itself
    assert r.lib.rtx_group_attach(r_same.ctx, None, C.byref(out)) == 1             # no base group
    assert r.lib.rtx_group_attach(None, r.group, C.byref(out)) == 1
    if (sc.width, sc.height) != (sc2.width, sc2.height):
        assert r.lib.rtx_group_attach(r_other.ctx, r.group, C.byref(out)) == 1     # another resolution
    r_same.group_attach(r)                                                         # the legitimate case still works afterwards
    r_same.group_render(); r.group_render()
    need = C.c_int32(0)
    assert r.lib.rtx_debug_blas_wide(r.ctx, 99, C.byref(need)) == 1
    assert r.lib.rtx_debug_blas_wide(r.ctx, 0, None) == 1
    r_same.group_destroy(); r.group_destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("knob", [{"RTX_DEBUG_FAIL_ITEM_ALLOC": "1"}, {"RTX_PK_ITEM_BYTES": str(5120 * 64 * 48)}, {"RTX_PK_ITEM_BYTES": "0"}])
def test_item_buffer_fallbacks_keep_the_frame_and_the_return_code(api, knob, monkeypatch):
    """The split shadow-ray walk's item buffer is bounded (RTX_PK_ITEM_BYTES; a full chunk makes the packet keep its nodes) and optional: when
    its allocation fails (forced here: a size no device has) the pending HIP error is cleared and the non-split kernel renders the frame —
    the call must return RTX_OK, rtx_get_stats too, and the frame must be the reference's."""
    for k, v in knob.items():
        monkeypatch.setenv(k, v)
    for name in ("materials_aniso", "tori16"):
        sc, g = util.load_golden(name)
        r = api.Renderer(sc)
        for mode in ({}, {"serial": True}):
            cmp = util.compare_to_golden(r.render(**mode), g)                 # render() raises on any non-zero status
            assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (knob, name, mode, cmp)


@pytest.mark.gpu
def test_invalid_knob_values_are_ignored(api, monkeypatch):
    """Knobs are parsed once in rtx_create, validated and clamped: zero / negative / non-numeric values keep the defaults (a zero-size
    k_shade grid or a negative shift count in the asm walkers would otherwise break production rendering through an inherited variable)."""
    for k, v in {"RTX_SHADE_GRID": "0", "RTX_PK_DEFER": "-5", "RTX_PK_GROW": "-3", "RTX_PK_SPLIT": "-64", "RTX_PK_DEFER_LEAF": "banana",
                 "RTX_SLOT_BUDGET": "-1", "RTX_ITEM_GRID": "0", "RTX_PK_GRID_ANY": "-2", "RTX_RESOLVE_BLOCK": "100", "RTX_LANE_FROM_LEVEL_ANY": "-7"}.items():
        monkeypatch.setenv(k, v)
    sc, g = util.load_golden("materials_aniso")
    cmp = util.compare_to_golden(api.Renderer(sc).render(), g)
    assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["packed_mismatch"] == 0, cmp


@pytest.mark.gpu
@pytest.mark.parametrize("n", [20, 40, 62])
def test_chain_shaped_tlas_takes_the_per_lane_kernels_instead_of_overflowing_the_packet_stack(api, n):
    """ONE 64-entry packet stack holds the TLAS part and the BLAS part of a walk, where the reference has a stack per BVH.  A degenerate
    chain-shaped TLAS (depth n - 1, legal for BVH_TRAVERSAL_STACK_SIZE 64) can need more than that: the host knows both depths and must hand
    such a scene to the per-lane kernels up front — right frame, RTX_OK from rtx_get_stats — instead of dropping pushes and reporting
    RTX_ERR_LIMIT after a wrong frame."""
    import orc
    from pyrtx import host, scene_io as sio
    sc, _ = util.load_golden("cube")
    root = sc.blas[0].nodes[0]
    inst = np.zeros(n, sio.INSTANCE); boxes = np.zeros((n, 6), np.float32)
    for k in range(n):                                               # cubes on a line receding from the camera, all visible through one another's gaps
        one, mn, mx = host.instance_update((0.9 * k - 2.0, 0.15 * k, 1.1 * k), (0.0, 0.0, 0.0, 1.0), root["aabb_min"], root["aabb_max"], 0)
        inst[k] = one[0]; boxes[k, :3] = mn; boxes[k, 3:] = mx
    nodes = np.zeros(2 * n, sio.BVH_NODE)                            # root 0, (unused 1), level d: pair (2d + 2, 2d + 3) = (leaf of instance d, rest of the chain)
    for d in range(n - 1):
        k = 0 if d == 0 else 2 * d + 1
        nodes[k]["aabb_min"] = boxes[d:, :3].min(axis=0); nodes[k]["aabb_max"] = boxes[d:, 3:].max(axis=0)
        nodes[k]["left_or_first"] = 2 * d + 2; nodes[k]["count"] = np.uint32((1 + d % 3) << 30).astype(np.int32)          # split axes vary: near / far orders differ per ray
        leaf = 2 * d + 2
        nodes[leaf]["aabb_min"] = boxes[d, :3]; nodes[leaf]["aabb_max"] = boxes[d, 3:]; nodes[leaf]["left_or_first"] = d; nodes[leaf]["count"] = 1
    last = 2 * (n - 1) + 1 if n > 1 else 0
    nodes[last]["aabb_min"] = boxes[n - 1, :3]; nodes[last]["aabb_max"] = boxes[n - 1, 3:]; nodes[last]["left_or_first"] = n - 1; nodes[last]["count"] = 1
    sc.instances, sc.tlas_nodes, sc.tlas_indices = inst, nodes, np.arange(n, dtype=np.int32)
    sc.config["bounces"] = 1
    r = api.Renderer(sc)
    ref = orc.OracleScene(sc).render(threads=8)
    for mode in ({}, {"serial": True}, {"packet_closest": True}):
        out = r.render(**mode)
        assert out["stats"] == ref["stats"] and np.array_equal(out["rgb"].view(np.uint32), ref["rgb"].view(np.uint32)), (n, mode)


@pytest.mark.parametrize("cam_y", ["1", "1.25", "-1", "0"])
@pytest.mark.parametrize("mesh", ["Cube", "Torus"])
def test_zero_direction_components_on_and_off_box_planes(api, mesh, cam_y):
    """A camera that is only translated has d.y == 0 on its middle row and d.x == 0 on its middle column: inverse components of +-inf.  The slab
    test then yields a NaN (0 * inf) only where the ray's origin lies exactly on a box plane of that axis; only those rays need the reference-form
    walker (csrc/rtx_packet.h pk_nan_possible: a search of the tree's distinct plane coordinates), all others — the common case — stay on the fast
    paths.  Camera heights on (1, -1: Cube.obj's faces; 0: planes of either tree) and off (1.25) the planes; frame == oracle in every launch shape."""
    import orc
    from pyrtx import assemble
    text = (f"size 64 48\nbounces 2\nmesh ./Data/{mesh}.obj 0 0 0\nmatset mesh:0:0 reflection 0.5 0.5 0.5\nplane 0 -1.5 0\n"
            f"point 20 20 20 0 {cam_y} -3\ndir 0.5 0.5 0.5 0 -1 0\ncamera 0 {cam_y} -4 0 0 0 1\n")
    sc = assemble.scene_from_script(text, os.path.join(util.GOLDEN, "meshes"), accel="sbvh")
    ref = orc.OracleScene(sc).render(threads=8)
    r = api.Renderer(sc)
    for mode in ({}, {"serial": True}, {"lane_trace": True}, {"packet_closest": True}, {"simple_trace": True}):
        out = r.render(**mode)
        assert out["stats"] == ref["stats"], mode
        assert np.array_equal(np.isnan(out["rgb"]), np.isnan(ref["rgb"])), mode
        assert util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"]), mode


@pytest.mark.parametrize("shape,n,ordered", [("chain", 20, 1), ("chain", 34, 0), ("chain", 60, 1), ("comb", 12, 1), ("comb", 24, 1)])
def test_deep_degenerate_blas_stacks_of_the_closest_hit_walk(api, shape, n, ordered):
    """Degenerate trees keep many entries pending on a lane's stack: past the 16 entries a lane holds in LDS the walk spills to HBM, with the
    pop-time keys (csrc/rtx_packet.h pk_lane_phase_closest).  "chain": n triangles in a row, every inner node = one leaf + the rest (depth n - 1;
    the ordered 4-wide records need about as many entries as the binary walk: 59 for n = 60).  "comb": every step = (leaf, leaf) + (leaf, rest):
    four slots per record, three of them pending — 1.5 entries per level, so that 24 steps (depth 47, legal for BVH_TRAVERSAL_STACK_SIZE 64) would
    need more than the 64-entry stack holds and the mesh keeps the binary walk (rtx_debug_blas_wide_closest == -1).  Frame == oracle in every
    launch shape, reflective triangles so that the deeper levels walk the tree from the inside too."""
    import copy
    import orc
    from pyrtx import scene_io as sio
    sc, _ = util.load_golden("cube")
    sc = copy.deepcopy(sc)
    n_tri = n if shape == "chain" else 3 * n + 1
    hot = np.zeros(n_tri, sio.TRI_HOT); cold = np.zeros(n_tri, sio.TRI_COLD)
    rng = np.random.default_rng(n)
    for i in range(n_tri):
        z = float(rng.uniform(-0.5, 0.5))
        hot["position_0"][i] = (i * 0.45 - n_tri * 0.225, -0.6, z); hot["position_edge_1"][i] = (0.4, 0.1 * (i % 3), 0.3 * ((i % 2) * 2 - 1)); hot["position_edge_2"][i] = (0.05, 1.2, 0.0)
        cold["normal_0"][i] = (0, 0, -1); cold["material_id"][i] = sc.blas[0].tri_cold["material_id"][0]
    # the tree as nested tuples over triangle indices in order, then laid out like the reference's builders: root 0, node 1 unused, children in pairs
    def chain(i): return i if i == n_tri - 1 else (i, chain(i + 1))
    def comb(i): return ((i, i + 1), (i + 2, comb(i + 3))) if i + 3 < n_tri else i
    import sys
    sys.setrecursionlimit(10000)
    tree = chain(0) if shape == "chain" else comb(0)
    nodes = np.zeros(2 * n_tri, sio.BVH_NODE)
    next_pair = [2]
    def place(t, idx):
        if isinstance(t, int):
            p = np.stack([hot["position_0"][t], hot["position_0"][t] + hot["position_edge_1"][t], hot["position_0"][t] + hot["position_edge_2"][t]]).astype(np.float32)
            nodes[idx]["aabb_min"] = p.min(axis=0); nodes[idx]["aabb_max"] = p.max(axis=0); nodes[idx]["left_or_first"] = t; nodes[idx]["count"] = 1
        else:
            l = next_pair[0]; next_pair[0] += 2
            place(t[0], l); place(t[1], l + 1)
            nodes[idx]["aabb_min"] = np.minimum(nodes[l]["aabb_min"], nodes[l + 1]["aabb_min"]); nodes[idx]["aabb_max"] = np.maximum(nodes[l]["aabb_max"], nodes[l + 1]["aabb_max"])
            nodes[idx]["left_or_first"] = l; nodes[idx]["count"] = 1 << 30          # split axis x
    place(tree, 0)
    sc.blas[0] = sio.Blas(nodes[:next_pair[0]].copy(), hot, cold, sc.blas[0].material_offset, 1)
    sc.materials["reflection"][:] = 0.4
    sc.config["traversal_strategy"] = ordered; sc.config["bounces"] = 2
    ref = orc.OracleScene(sc).render(threads=8)
    r = api.Renderer(sc)
    need = r.debug_blas_wide_closest(0)
    assert need == {("chain", 20): 19, ("chain", 34): 33, ("chain", 60): 59, ("comb", 12): 36, ("comb", 24): -1}[(shape, n)], need
    for mode in ({}, {"serial": True}, {"lane_trace": True}, {"packet_closest": True}):
        out = r.render(**mode)
        assert out["stats"] == ref["stats"], mode
        assert util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"]), mode
    assert int(ref["stats"]["reflection"]) > 0


def test_levels_that_cannot_hold_rays_are_not_launched(api):
    """A hit spawns a reflection / refraction ray only where its material's Ks / Kt is not all zero (Raytracer.cpp:204-213).  With no such material
    uploaded the levels >= 1 are provably empty whatever NUMBER_OF_BOUNCES says, and rtx_render_tiles queues one level's launches instead of four
    levels' (three launches of 7-9 us each per level).  Same frame, same ray counts; with one reflective material back, all levels are launched."""
    import copy
    import orc
    sc, _ = util.load_golden("monkey_small")
    sc = copy.deepcopy(sc)
    assert int(sc.config["bounces"][0] if hasattr(sc.config["bounces"], "__len__") else sc.config["bounces"]) == 3
    launches = {}
    for tag, ks in (("diffuse", 0.0), ("mirror", 0.25)):
        sc.materials["reflection"][:] = ks; sc.materials["transmittance"][:] = 0.0
        ref = orc.OracleScene(sc).render(threads=8)
        r = api.Renderer(sc)
        r.render(serial=True)
        r.enable_timing(True)
        out = r.render(serial=True)
        names = [k for k, _ in r.kernel_times()]
        r.enable_timing(False)
        launches[tag] = names
        assert out["stats"] == ref["stats"] and util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"]), tag
        for mode in ({}, {"lane_trace": True}, {"simple_trace": True}):
            o2 = api.Renderer(sc).render(**mode)
            assert o2["stats"] == ref["stats"] and util.bit_exact(o2["rgb"], ref["rgb"]), (tag, mode)
    assert launches["diffuse"].count("k_shade") == 1 and launches["diffuse"].count("k_resolve") == 1 and launches["diffuse"].count("k_trace_closest") == 1, launches["diffuse"]
    assert launches["mirror"].count("k_shade") == 4 and launches["mirror"].count("k_resolve") == 4 and launches["mirror"].count("k_trace_closest") == 4, launches["mirror"]


@pytest.mark.parametrize("name", ["materials_aniso", "dynamic", "monkey_small"])
def test_graph_replay_of_identical_calls_changes_nothing(api, name, monkeypatch):
    """RTX_GRAPH=1: the second identical rtx_render_tiles call (same scene, queues, tile range, flags) is captured into a hipGraph, later ones replay it
    (host cost per frame 40-58 -> 9-21 us; the GPU side is the same launches).  A call that differs in anything the kernels' arguments depend on runs
    eagerly again and re-captures.  Every frame — eager, captured, replayed, after a change of flags and back — is the reference's."""
    monkeypatch.setenv("RTX_GRAPH", "1")
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    for mode in ({"serial": True}, {}, {"serial": True, "cull_dead_shadow_rays": True}, {"serial": True}):
        for i in range(4):
            cmp = util.compare_to_golden(r.render(**mode), g)
            assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (mode, i, cmp)
    half = sc.tile_count // 2
    for i in range(3):                      # a different tile range: its own capture; the other half of the frame keeps the earlier pixels
        r.render_async(0, 1, half, serial=True)
    r.synchronize()
    _, packed = r.framebuffer()
    assert np.array_equal(packed, np.asarray(g["packed"]).reshape(packed.shape))
