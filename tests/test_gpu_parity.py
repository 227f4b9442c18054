"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle and the
committed golden vectors produced by the real reference (tests/golden, see
oracle/ref_harness/make_goldens.py).

Tolerance: BASELINE.json states per-pixel RGB max-abs error < 1e-4.  What is asserted is stricter: every
frame is BIT-EXACT (fp32 colour of every pixel, packed pixel, ray counts) — the arithmetic is unfused fp32 in the
reference's evaluation order, and acosf / atan2f / expf / log2f are glibc's algorithms bit for bit (csrc/rtx_libm.h).
"""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def api():
    from pyrtx import api as a
    a.load_library()
    return a


# every launch shape / kernel selection rtx_render_tiles knows (include/rtx.h): the default two-stream shape, the one-stream shape
# bench.py times (RTX_RENDER_SERIAL: ONE merged shadow-ray launch for all levels), and the A/B kernel choices
MODES = {
    "default": {},
    "serial": {"serial": True},
    "lane": {"lane_trace": True},
    "serial_lane": {"serial": True, "lane_trace": True},
    "serial_packet_closest": {"serial": True, "packet_closest": True},
    "packet_closest": {"packet_closest": True},
    "serial_cull": {"serial": True, "cull_dead_shadow_rays": True},
}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", list(util.GOLDENS))
def test_frame_matches_reference_golden(api, name, mode):
    sc, g = util.load_golden(name)
    out = api.Renderer(sc).render(**MODES[mode])
    cmp = util.compare_to_golden(out, g)
    assert cmp["stats_equal"], (out["stats"], g["stats"])
    assert cmp["nan_mismatch"] == 0
    assert cmp["max_abs"] < TOL, cmp
    assert cmp["packed_mismatch"] == 0, cmp
    assert cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0, cmp         # not merely within tolerance: identical


@pytest.mark.parametrize("name", ["cube", "monkey_small", "materials_aniso", "dynamic", "tori16"])
def test_frame_matches_oracle_and_work_counters(api, name):
    import orc
    sc, g = util.load_golden(name)
    out = api.Renderer(sc).render(count_work=True)
    ref = orc.OracleScene(sc).render(threads=8)
    assert out["stats"] == ref["stats"]
    d = np.abs(out["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
    assert np.nanmax(d) < TOL and util.bit_exact(out["rgb"], ref["rgb"])
    assert np.array_equal(out["packed"], ref["packed"])
    # the roofline accounting relies on these agreeing exactly with the CPU restatement
    for k in ("closest_rays", "any_rays", "tlas_nodes_closest", "tlas_nodes_any", "blas_nodes_closest", "blas_nodes_any",
              "instances_closest", "instances_any", "tri_tests_closest", "tri_tests_any", "shaded_hits", "sky_lookups",
              "texel_fetches", "rays_spawned"):
        assert out["work"][k] == ref["work"][k], (k, out["work"][k], ref["work"][k])


def test_scenes_without_transcendentals_are_bit_exact(api):
    for name in ("cube", "monkey_small"):
        sc, g = util.load_golden(name)
        out = api.Renderer(sc).render()
        assert np.array_equal(out["rgb"].view(np.uint32), g["rgb"].view(np.uint32)), name


def test_tile_subsets_and_strides(api):
    """Round-robin tile shards (the multi-GPU partition) reproduce the full frame."""
    sc, g = util.load_golden("materials_aniso")
    full = api.Renderer(sc).render()
    r = api.Renderer(sc)
    for shard in range(3):
        n = (sc.tile_count - shard + 2) // 3
        r.render_async(first_tile=shard, tile_stride=3, tile_count=n)
    rgb, packed = r.framebuffer()
    assert np.array_equal(rgb.view(np.uint32), full["rgb"].view(np.uint32))
    assert np.array_equal(packed, full["packed"])
    # a single tile only touches its own pixels
    r2 = api.Renderer(sc)
    r2.render_async(first_tile=5, tile_stride=1, tile_count=1)
    rgb2, _ = r2.framebuffer()
    ty, tx = (5 // sc.tile_count_x) * 32, (5 % sc.tile_count_x) * 32
    mask = np.zeros((sc.height, sc.width), bool); mask[ty:ty + 32, tx:tx + 32] = True
    assert np.all(rgb2[~mask] == 0)
    assert np.array_equal(rgb2[mask].view(np.uint32), full["rgb"][mask].view(np.uint32))


def test_empty_and_invalid_inputs(api):
    sc, _ = util.load_golden("cube")
    r = api.Renderer(sc)
    r.render_async(first_tile=0, tile_stride=1, tile_count=0)          # empty batch is legal
    st, _ = r.stats()
    assert st == {"primary": 0, "shadow": 0, "reflection": 0, "refraction": 0}
    with pytest.raises(api.RtxError):
        r.render_async(first_tile=sc.tile_count, tile_stride=1, tile_count=1)   # out of range tile
    with pytest.raises(api.RtxError):
        r.render_async(first_tile=0, tile_stride=0, tile_count=1)
    # render before set_frame
    r3 = api.Renderer(sc, upload=False)
    with pytest.raises(api.RtxError) as e:
        r3.render_async()
    assert e.value.code == 5
    # MAX_MATERIALS (Config.h:18)
    import ctypes as C
    mats = np.zeros(257, util.sio.MATERIAL)
    assert r3.lib.rtx_upload_materials(r3.ctx, mats.ctypes.data, 257) == 4


def test_ragged_resolution(api):
    """Width/height not multiples of 32: edge tiles are clipped (WorkerThread.cpp:60-61)."""
    import orc
    sc, _ = util.load_golden("monkey_small")
    sc.config["width"] = 301; sc.config["height"] = 173
    # camera basis depends on the resolution (Camera.cpp:5-16): rebuild top-left for identity-free scenes is not needed
    out = api.Renderer(sc).render()
    ref = orc.OracleScene(sc).render(threads=8)
    assert np.array_equal(out["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    assert out["stats"] == ref["stats"]


@pytest.mark.parametrize("name", ["materials_aniso", "dynamic", "tori16"])
def test_pair_fetch_kernels_equal_plain_traversal(api, name):
    """k_trace_fast (child-pair fetch + early filter) must reproduce the plain pop-and-test kernels bit for bit."""
    sc, _ = util.load_golden(name)
    r = api.Renderer(sc)
    a = r.render()
    b = r.render(simple_trace=True)
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert a["stats"] == b["stats"]


@pytest.mark.parametrize("mode", ["default", "serial", "serial_packet_closest"])
def test_atrium_cfg3_small_matches_oracle(api, mode):
    """The cfg3 workload (own BVH builder, 64k-triangle detail level) at reduced resolution vs the oracle."""
    import orc
    from pyrtx import host
    sc = host.atrium_scene(480, 270, 3, detail=0)
    out = api.Renderer(sc).render(**MODES[mode])
    ref = orc.OracleScene(sc).render(threads=8)
    assert out["stats"] == ref["stats"]
    d = np.abs(out["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
    assert np.nanmax(d) < TOL and util.bit_exact(out["rgb"], ref["rgb"])
    assert np.array_equal(out["packed"], ref["packed"])


@pytest.mark.parametrize("name", ["materials_aniso", "dynamic", "tori16", "monkey_small"])
def test_culling_dead_shadow_rays_changes_nothing(api, name):
    """RTX_RENDER_CULL_DEAD_SHADOW_RAYS: frame and PerformanceStats are bit-identical with the flag on."""
    sc, _ = util.load_golden(name)
    r = api.Renderer(sc)
    a = r.render()
    b = r.render(cull_dead_shadow_rays=True)
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert np.array_equal(a["packed"], b["packed"]) and a["stats"] == b["stats"]


def test_bench_gather_path_on_one_gpu():
    """bench.py's N>1 plumbing (tile shard -> index_select -> RCCL gather -> un-swizzle) rehearsed with one rank over nccl."""
    import json, os, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(util.REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "materials",
                          "--cpu-seconds", "0", "--force-gather"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["gathered_frame_equals_single_gpu_frame"] is True
    assert j["rays_per_frame"]["total"] == 57600 + 191709 + 52955 + 12692


@pytest.mark.parametrize("name", ["monkey_small_heat", "materials_heat"])
def test_heatmap_mode_is_bit_exact(api, name):
    """BVH_VISUALIZE_HEATMAP: integer step counts times powers of two, so the frame must equal the reference's bit for bit;
    only primary rays are counted; rendering the odd and even tiles separately composes the same frame."""
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    out = r.render()
    cmp = util.compare_to_golden(out, g)
    assert cmp["bit_exact"] and cmp["packed_mismatch"] == 0 and cmp["stats_equal"], cmp
    assert out["stats"]["shadow"] == 0 and out["stats"]["reflection"] == 0 and out["stats"]["refraction"] == 0
    counted = r.render(count_work=True)
    assert np.array_equal(counted["rgb"], out["rgb"]) and counted["work"]["closest_rays"] == sc.width * sc.height
    r2 = api.Renderer(sc)
    T = sc.tile_count
    r2.render(first_tile=0, tile_stride=2, tile_count=(T + 1) // 2)
    half = r2.render(first_tile=1, tile_stride=2, tile_count=T // 2)
    assert np.array_equal(half["packed"], out["packed"])


def test_three_contexts_in_flight_serial_mode(api):
    """What bench.py times: three contexts, each on its own stream, rendering DIFFERENT frames concurrently with RTX_RENDER_SERIAL
    (cfg5 frames 1-3: per-frame TLAS), every frame compared with the oracle after all three were queued without a host sync."""
    import orc
    from test_dynamic_frames import frame_state
    scenes, renderers = [], []
    for f in (1, 2, 3):
        sc, g, inst, nodes, idx = frame_state("tori16", f)
        sc.instances, sc.tlas_nodes, sc.tlas_indices = inst, nodes, idx
        scenes.append(sc); renderers.append(api.Renderer(sc))
    for rounds in range(3):                                     # keep the three streams busy for a while
        for r in renderers:
            r.render_async(serial=True)
    outs = []
    for r in renderers:
        st, _ = r.stats(); rgb, packed = r.framebuffer()
        outs.append({"rgb": rgb, "packed": packed, "stats": st})
    for sc, out in zip(scenes, outs):
        ref = orc.OracleScene(sc).render(threads=8)
        assert out["stats"] == ref["stats"]
        assert util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"])


def test_stack_size_knob_is_honoured(api):
    """BVH_TRAVERSAL_STACK_SIZE (Config.h:25): the reference's per-BVH stack holds one pending sibling per ancestor entered through its
    near child and two more while a node is expanded; a configured size the scene's deepest expansion does not fit in must be reported
    (RTX_ERR_LIMIT from rtx_get_stats), a sufficient one must not."""
    sc, _ = util.load_golden("monkey_small")
    sc.config["stack_size"] = 64
    api.Renderer(sc).render()                                    # fits
    sc.config["stack_size"] = 3
    for mode in ("default", "packet_closest", "lane"):
        with pytest.raises(api.RtxError) as e:
            api.Renderer(sc).render(**MODES[mode])
        assert e.value.code == 4, mode


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("name", ["materials_aniso", "monkey_small"])
def test_group_path_loopback_equals_single_gpu_frame(api, name, world):
    """include/rtx.h rtx_group_*: tile t -> rank t mod world, tile-major packed buffers, frame assembly on rank 0 — replayed for `world`
    ranks on one GPU (rtx_debug_group_loopback: everything but the ncclGather itself).  monkey_small is 301-pixel-ragged below."""
    sc, _ = util.load_golden(name)
    if name == "monkey_small":
        sc.config["width"] = 301; sc.config["height"] = 173            # edge tiles clipped in both directions
    full = api.Renderer(sc).render()
    r = api.Renderer(sc)
    r.group_loopback(world)
    _, packed = r.framebuffer()
    assert np.array_equal(packed, full["packed"])


def test_group_render_one_rank_over_rccl(api):
    """rtx_group_create / rtx_group_render with world == 1: the real RCCL communicator and ncclGather (to itself), through the C ABI."""
    sc, _ = util.load_golden("materials_aniso")
    full = api.Renderer(sc).render()
    r = api.Renderer(sc)
    r.group_create(0, 1, api.Renderer.group_unique_id())
    for _ in range(2):
        r.group_render()
    st, _ = r.stats()
    _, packed = r.framebuffer()
    assert np.array_equal(packed, full["packed"]) and st == full["stats"]
    r.group_destroy()


def test_group_frames_in_flight_share_one_communicator(api):
    """bench.py's N > 1 shape on one GPU: three contexts (frames in flight) of one rank, the second and third JOIN the first one's
    communicator (rtx_group_attach); their gathers go through the rank's exchange stream in call order.  Different frames per context
    (cfg5 frames 0 and 1), several rounds without host synchronisation in between; each assembled frame must be the single-GPU frame."""
    scenes = [util.load_golden(n)[0] for n in ("tori16", "tori16_f1", "tori16")]
    full = [api.Renderer(s).render() for s in scenes]
    rs = [api.Renderer(s) for s in scenes]
    rs[0].group_create(0, 1, api.Renderer.group_unique_id())
    rs[1].group_attach(rs[0]); rs[2].group_attach(rs[0])
    for _ in range(4):
        for r in rs:
            r.group_render()
    for r, f in zip(rs, full):
        st, _ = r.stats()
        _, packed = r.framebuffer()
        assert np.array_equal(packed, f["packed"]) and st == f["stats"]
    for r in (rs[1], rs[0], rs[2]):               # any destruction order: the communicator goes with its last member
        r.group_destroy()


def test_shadow_walk_uses_wide_records_for_reference_trees_and_binary_walk_otherwise(api):
    """The shadow-ray packet walk takes the 4-wide records (csrc/rtx_api.hip build_nodes_pk4) only for meshes whose child boxes lie
    inside their parents' boxes — every tree the reference builds.  A tree that violates the nesting keeps the binary walk, and both
    produce the frame of the oracle, which walks the binary tree node by node as BottomLevelBVH::intersect does."""
    import copy
    import orc
    for name in ["monkey_small", "tori16", "materials_aniso"]:
        sc, g = util.load_golden(name)
        r = api.Renderer(sc)
        needs = [r.debug_blas_wide(i) for i in range(len(sc.blas))]
        assert all(0 <= n <= 36 for n in needs), (name, needs)
        needs_c = [r.debug_blas_wide_closest(i) for i in range(len(sc.blas))]      # the closest-hit per-lane walk's ordered records (build_nodes_pk4c)
        assert all(0 <= n <= 62 for n in needs_c), (name, needs_c)
    sc, g = util.load_golden("monkey_small")
    sc2 = copy.deepcopy(sc)
    rng = np.random.default_rng(5)
    for b in sc2.blas:
        nodes = b.nodes.copy()
        # grow the boxes of some non-root nodes beyond their parents' (the tree stays a valid, conservative BVH; only nesting is lost)
        inner = np.nonzero((nodes["count"] & 0x3FFFFFFF) == 0)[0]
        picks = rng.choice(inner[inner > 0], size=min(40, len(inner) - 1), replace=False) if len(inner) > 1 else []
        for i in picks:
            l = int(nodes["left_or_first"][i])
            nodes["aabb_min"][l] -= np.float32(0.37); nodes["aabb_max"][l + 1] += np.float32(0.21)
        b.nodes = nodes
    r2 = api.Renderer(sc2)
    assert all(r2.debug_blas_wide(i) == -1 and r2.debug_blas_wide_closest(i) == -1 for i in range(len(sc2.blas)))
    ref = orc.OracleScene(sc2).render(threads=8)
    for mode in ({}, {"serial": True}, {"lane_trace": True}):
        out = r2.render(**mode)
        assert out["stats"] == ref["stats"]
        assert util.bit_exact(out["rgb"], ref["rgb"]) and np.array_equal(out["packed"], ref["packed"])
    # the same mesh with its original, nested boxes renders through the wide records and matches the reference's golden frame
    out = api.Renderer(sc).render()
    cmp = util.compare_to_golden(out, g)
    assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["packed_mismatch"] == 0


@pytest.mark.parametrize("lane_from", ["99", "1", "3"])
@pytest.mark.parametrize("name", ["dynamic", "tori16", "materials_aniso"])
def test_per_level_kernel_selection_changes_nothing(api, name, lane_from, monkeypatch):
    """Multi-instance scenes trace levels >= 2 with the per-lane kernels by default (csrc/rtx_api.hip, `lane_from`); whatever the split —
    packets at every level (99), per-lane from level 1 or 3 — the frame is the reference's, in both launch shapes."""
    monkeypatch.setenv("RTX_LANE_FROM_LEVEL", lane_from)
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    for mode in ({}, {"serial": True}):
        cmp = util.compare_to_golden(r.render(**mode), g)
        assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (mode, cmp)


@pytest.mark.parametrize("split", ["0", "2048", "64"])
@pytest.mark.parametrize("name", ["monkey_small", "materials_aniso", "dynamic", "tori16"])
def test_split_shadow_walk_changes_nothing(api, name, split, monkeypatch):
    """RTX_PK_SPLIT: the shadow-ray packet kernel hands the nodes few lanes want to a second kernel as 48-byte items (csrc/rtx_packet.h
    k_items) instead of walking them in its own per-lane phase.  Off, on, and on with chunks so small (64 items) that most hand-overs are
    refused and the packet keeps the node: always the reference's frame, in both launch shapes."""
    monkeypatch.setenv("RTX_PK_SPLIT", split)
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    for mode in ({}, {"serial": True}, {"serial": True, "cull_dead_shadow_rays": True}):
        cmp = util.compare_to_golden(r.render(**mode), g)
        assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (mode, cmp)


@pytest.mark.parametrize("knobs", [{"RTX_FUSE_SHADE": "1"}, {"RTX_PK_CLOSEST_ASM": "0", "RTX_PK_DEFER_PRIMARY": "8"}, {"RTX_PK_DEFER_PRIMARY": "64"},
                                   {"RTX_PK_DEFER_PRIMARY": "1", "RTX_PK_DEFER_CLOSEST": "2"}, {"RTX_PK_DEFER_PRIMARY": "16", "RTX_PK_DEFER_CLOSEST": "16", "RTX_LANE_FROM_LEVEL": "99"},
                                   {"RTX_PK_WIDE_CLOSEST": "0"}, {"RTX_PK_WIDE_CLOSEST": "0", "RTX_PK_DEFER_PRIMARY": "64", "RTX_LANE_FROM_LEVEL": "99"},
                                   {"RTX_PK_CLOSEST_ASM": "0", "RTX_PK_DEFER_PRIMARY": "8", "RTX_PK_WIDE_CLOSEST": "0"}],
                         ids=["fused_shade", "compiled_shared_walk", "all_private", "share_everything", "share_all_levels_packets_everywhere",
                              "binary_private_walk", "binary_private_walk_all_private_packets_everywhere", "compiled_shared_walk_binary_private_walk"])
@pytest.mark.parametrize("name", ["materials_aniso", "dynamic", "tori16", "materials_b5", "materials_naive"])
def test_closest_hit_walk_variants_change_nothing(api, name, knobs, monkeypatch):
    """Round 3's closest-hit choices are scheduling decisions: the hand-scheduled shared walk (pk_blas_closest_asm) with any hand-over
    threshold, for primary packets only or for every level, the compiled shared walk it replaced (RTX_PK_CLOSEST_ASM=0), every lane private,
    the opt-in fused trace-and-shade kernel (RTX_FUSE_SHADE=1: child slots from wave-private chunks, unused slots marked as holding no
    ray), and the private lanes' walk over the ordered 4-wide records (default) or the binary tree (RTX_PK_WIDE_CLOSEST=0) — always the
    reference's frame and ray counts, in both launch shapes."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    for mode in ({}, {"serial": True}, {"serial": True, "cull_dead_shadow_rays": True}):
        cmp = util.compare_to_golden(r.render(**mode), g)
        assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (knobs, mode, cmp)


@pytest.mark.parametrize("lpt", ["1", "-1", "0"])
@pytest.mark.parametrize("name", ["materials_aniso", "dynamic", "tori16", "monkey_small"])
def test_longest_packets_first_changes_nothing(api, name, lpt, monkeypatch):
    """RTX_PK_LPT: from the second call over the same tiles on, the level-0 closest-hit launch takes its packets longest first, by what each packet
    cost in the previous call (costs written by the launch, sorted on a side stream: csrc/rtx_api.hip k_packet_order).  A schedule, not a result:
    every frame of a series — the first (queue order), the later ones (sorted), after a change of tile range and back — is the reference's, in both
    launch shapes; 1 = always, -1 = the default rule (two-stream shape, at least two packets per wave), 0 = never."""
    monkeypatch.setenv("RTX_PK_LPT", lpt)
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    for mode in ({}, {"serial": True}, {}):
        for i in range(4):
            cmp = util.compare_to_golden(r.render(**mode), g)
            assert cmp["stats_equal"] and cmp["max_abs"] == 0.0 and cmp["n_diff_pixels"] == 0 and cmp["packed_mismatch"] == 0, (mode, i, cmp)
    half = sc.tile_count // 2
    for i in range(3):                      # another tile range: its own history; the rest of the frame keeps the earlier pixels
        r.render_async(0, 1, half)
    for i in range(3):
        r.render_async(half, 1, sc.tile_count - half)
    r.synchronize()
    _, packed = r.framebuffer()
    assert np.array_equal(packed, np.asarray(g["packed"]).reshape(packed.shape))
