// rtx_api.hip — host side of librtx_hip.so: the C ABI of include/rtx.h over the gfx950 kernels.
//
// One context = one GPU, one HIP stream.  Scene data is uploaded once and stays resident in HBM;
// rtx_set_frame uploads what Scene::update produces (< 8 KiB for 16 instances); rtx_render_tiles
// enqueues the whole wavefront pipeline for a batch of tiles without any host synchronisation:
//
//   for level = 0 .. bounces:  k_trace<closest>  ->  k_shade  ->  k_trace<any>
//   for level = bounces .. 0:  k_resolve
//
// Queue sizes are only known on the device (DevCounters), so every kernel reads its element count
// from HBM and the launch geometry is fixed (persistent grids).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/rtx.h"
#include "rtx_device.h"
#include "rtx_math.h"
#include "rtx_libm.h"
#include "rtx_texture.h"
#include "rtx_trace.h"
#include "rtx_shade.h"
#include "rtx_packet.h"
#include "rtx_present.h"

#define SLOT_BUDGET (48ll * 1000 * 1000)      // ray slots per batch of tiles (see plan_batch)

struct DevBuf {
    void * p = nullptr; size_t cap = 0;
};

// Tuning knobs (environment, A/B runs and tests): read ONCE per context in rtx_create, validated and clamped there; a value that does not
// parse or lies outside its range leaves the default in place.  Nothing on the render path calls getenv.
static long long knob_int(const char * name, long long dflt, long long lo, long long hi) {
    const char * e = getenv(name);
    if (!e || !*e) return dflt;
    char * end = nullptr; const long long v = strtoll(e, &end, 10);
    if (end == e || *end != 0 || v < lo || v > hi) { fprintf(stderr, "librtx_hip: %s=%s ignored (expected an integer in [%lld, %lld])\n", name, e, lo, hi); return dflt; }
    return v;
}
static double knob_real(const char * name, double dflt, double lo, double hi) {
    const char * e = getenv(name);
    if (!e || !*e) return dflt;
    char * end = nullptr; const double v = strtod(e, &end);
    if (end == e || *end != 0 || !(v >= lo && v <= hi)) { fprintf(stderr, "librtx_hip: %s=%s ignored (expected a number in [%g, %g])\n", name, e, lo, hi); return dflt; }
    return v;
}
struct Knobs {
    long long slot_budget;       // RTX_SLOT_BUDGET        ray slots per batch of tiles (plan_batch)
    long long item_bytes_max;    // RTX_PK_ITEM_BYTES      upper bound of the split walk's item buffer per context
    int shade_grid;              // RTX_SHADE_GRID         k_shade workgroups per CU
    int lane_from_level;         // RTX_LANE_FROM_LEVEL    per-lane kernels from this level on (-1: the rule in render_tiles_impl)
    int lane_from_level_any;     // RTX_LANE_FROM_LEVEL_ANY  the same for shadow rays only (-1: follow the rule)
    int split_items;             // RTX_PK_SPLIT           split shadow-ray walk: 0 = off, else the smallest item chunk
    int resolve_block;           // RTX_RESOLVE_BLOCK      k_resolve workgroup size
    int pk4_order;               // RTX_PK4_ORDER          slot order of the 4-wide records (0: smallest stack need first)
    bool merge_any, no_wide, no_wide_closest, fail_item_alloc, graph; int lpt;
    bool fuse_shade;             // RTX_FUSE_SHADE         the closest-hit packet kernel shades its own hits (0: a k_shade launch per level)
};

struct KernelTime { const char * name; hipEvent_t a, b; };

struct rtx_ctx {
    rtx_config cfg;
    hipStream_t stream = nullptr;        // stream all work is enqueued on
    hipStream_t own_stream = nullptr;    // created by rtx_create
    hipStream_t any_stream = nullptr;    // shadow-ray kernels run here, overlapping the next levels' closest-hit / shade kernels
    hipEvent_t ev_shade0 = nullptr, ev_shade_last = nullptr, ev_any_done = nullptr;
    void * ext_rgb = nullptr, * ext_packed = nullptr;
    std::string err;
    int n_cu = 0;

    std::vector<DevBlas> h_blas;
    std::vector<std::vector<void *>> blas_allocs;
    DevBuf d_blas, d_materials, d_textures, d_sky, d_ewa;
    std::vector<DevTexture> h_tex;
    std::vector<void *> tex_allocs;
    int material_count = 0, sky_size = 0;
    // host copies for validate_references(): every id a kernel will follow is checked on the host before anything is launched
    std::vector<rtx_material> h_materials; std::vector<int> blas_max_local_material, blas_inner_depth, blas_packet_ok; int tlas_inner_depth = -1; std::vector<int> frame_primitive_materials; bool refs_dirty = true; Knobs knobs; int item_blocks = 0, item_cap_alloc = 0; DevBuf d_pk_items, d_pk_item_count;

    // per-frame state (rtx_set_frame): ONE device block, filled by one asynchronous copy on the context's stream from a ring of pinned
    // staging buffers — the host never waits for the GPU between frames (cfg5: Scene::update + TLAS rebuild every frame)
    DevBuf d_frame;
    struct FrameStage { void * host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; } stage[3];
    int stage_next = 0;
    DevScene scene;
    bool frame_set = false;

    DevQueues q;
    DevBuf qb[20];
    size_t slots_alloc = 0, shadow_alloc = 0;
    DevBuf d_dbg_s0, d_dbg_s1, d_stats_partial, d_pk_fifo, d_counters, d_spill, d_fb_rgb, d_fb_packed, d_display, d_gamma, d_pk_heads;
    int trace_blocks_closest = 0, trace_blocks_any = 0, trace_blocks_count = 0;
    int pk_blocks_closest = 0, pk_blocks_any = 0;    // persistent grids of the packet kernels (even: waves % 8 == 0, see k_begin_batch)

    rtx_stats stats_acc; rtx_work_counters work_acc; uint32_t err_flags_acc = 0;
    bool stats_pending = false;
    std::vector<DevCounters *> pending_host;   // pinned copies, one per batch

    bool timing = false;
    bool serial = false;                 // RTX_RENDER_SERIAL: keep every kernel on one stream (per-kernel timings without overlap)
    // RTX_GRAPH=1: the launches of a rtx_render_tiles call are captured once into a hipGraph and replayed while nothing they depend on changes
    hipGraphExec_t graph_exec = nullptr; std::vector<unsigned char> graph_key, graph_warm;
    // RTX_PK_LPT: level-0 closest-hit packets longest first, by their cost in the previous call with the same tiles (k_packet_order on a side stream)
    DevBuf d_pk_cost, d_pk_order; hipStream_t order_stream = nullptr; hipEvent_t ev_cost = nullptr, ev_order = nullptr; int32_t lpt_key[4] = { -1, -1, -1, -1 }; bool lpt_valid = false;
    std::vector<KernelTime> times;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
};

#define HIP_OK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return RTX_ERR_HIP; } } while (0)

static int ensure(rtx_ctx * c, DevBuf & b, size_t bytes) {
    if (bytes <= b.cap && b.p) return RTX_OK;
    if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { c->err = std::string("hipMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP; }
    b.cap = bytes;
    return RTX_OK;
}

static int upload(rtx_ctx * c, DevBuf & b, const void * src, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_OK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RTX_OK;
}

static void convert_nodes(const rtx_bvh_node * nodes, int n, std::vector<float4> & out) {
    out.resize((size_t)2 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        float l, cnt;
        memcpy(&l, &nodes[i].left_or_first, 4); memcpy(&cnt, &nodes[i].count, 4);
        out[2 * i]     = make_float4(nodes[i].aabb_min[0], nodes[i].aabb_min[1], nodes[i].aabb_min[2], l);
        out[2 * i + 1] = make_float4(nodes[i].aabb_max[0], nodes[i].aabb_max[1], nodes[i].aabb_max[2], cnt);
    }
}

// the packet kernels' node record: every SGPR pair of a fetched node is one operand of a packed-fp32 instruction (rtx_packet.h)
static void convert_nodes_pk(const rtx_bvh_node * nodes, int n, std::vector<float4> & out) {
    out.resize((size_t)2 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        float l, cnt;
        memcpy(&l, &nodes[i].left_or_first, 4); memcpy(&cnt, &nodes[i].count, 4);
        out[2 * i]     = make_float4(nodes[i].aabb_min[0], nodes[i].aabb_min[1], nodes[i].aabb_max[0], nodes[i].aabb_max[1]);
        out[2 * i + 1] = make_float4(nodes[i].aabb_min[2], nodes[i].aabb_max[2], l, cnt);
    }
}

// The shadow-ray packet walk's 4-wide node records (rtx_packet.h, pk_blas_any_asm4).  BottomLevelBVH::intersect (BottomLevelBVH.cpp:398-437)
// answers "is any triangle hit": the answer does not depend on the order the tree is walked in, and when every child box lies inside its
// parent's box — componentwise, in the stored floats — it does not depend on the inner nodes either: AABB::intersect (AABB.cpp:38-52) is
// monotone under nesting in floating point (x -> fl(x - o) and x -> fl(x * inv) are monotone, so t_near(parent) <= t_near(child) and
// t_far(child) <= t_far(parent) hold exactly, and a ray that passes a child's strict test passes its parent's).  The set of leaves a ray
// reaches is therefore the set of leaves whose own box it passes together with all ancestors' = (under nesting) the leaves reached through
// ANY subset of the ancestors.  The record of inner node j lists its grandchildren (children where a child is a leaf): the packet walk
// tests 4 boxes per fetch and takes half the dependent steps.  Trees that are not nested (or exceed the packed-entry limits) keep the
// binary walk.  Slot order = visiting order: largest box first (the likeliest occluder), unless that would need too many packet-stack entries
// at this node — then smallest stack need first, which bounds the occupancy by the tree's Strahler-like number.  *stack_need = the bound.
//   record (128 B, at byte offset 64 * left-child index of j): 4 x { min.x, min.y, max.x, max.y, min.z, max.z, first, leaf count };
//   first = left-child index of the slot node (inner, count 0) or first triangle (leaf); unused slots hold a point box, which never passes.
static bool build_nodes_pk4(const rtx_bvh_node * nodes, int n, int tri_count, std::vector<float4> & out, int * stack_need, int order_mode) {
    if (n >= (1 << 24) || tri_count >= (1 << 24)) return false;
    auto inside = [&](int c, int p) {
        for (int a = 0; a < 3; a++) if (!(nodes[c].aabb_min[a] >= nodes[p].aabb_min[a] && nodes[c].aabb_max[a] <= nodes[p].aabb_max[a])) return false;
        return true;
    };
    // reachable inner nodes in pre-order (validate_tree has already checked the structure)
    std::vector<int> order, stack(1, 0);
    while (!stack.empty()) {
        const int i = stack.back(); stack.pop_back();
        const int cnt = nodes[i].count & 0x3fffffff, f = nodes[i].left_or_first;
        for (int a = 0; a < 3; a++) if (!(nodes[i].aabb_min[a] <= nodes[i].aabb_max[a])) return false;      // the sign-coherent slab test relies on min <= max
        if (cnt > 0) { if (cnt >= 256) return false; continue; }
        if (f & 1) return false;                                   // records are addressed by left / 2
        if (!inside(f, i) || !inside(f + 1, i)) return false;      // nesting
        order.push_back(i); stack.push_back(f); stack.push_back(f + 1);
    }
    out.assign((size_t)4 * (n > 0 ? n : 1) + 8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<int> need((size_t)n, 0);                            // packet-stack entries the walk of a subtree can have pending
    // visiting order = slot order.  For a shadow ray any hit ends the walk, so the likeliest occluder should come first: largest box first
    // (merged launch 0.652 -> 0.627 ms; by triangle count, density or an RTSAH-style area^2 / cost key: 0.635-0.70).  Where that order would
    // need more packet-stack entries than the cap, the node falls back to the order that minimises the need (smallest need first).
    // order_mode (RTX_PK4_ORDER) 0: always smallest need first (A/B runs)
    auto area = [&](int i) { const double dx = (double)nodes[i].aabb_max[0] - nodes[i].aabb_min[0], dy = (double)nodes[i].aabb_max[1] - nodes[i].aabb_min[1], dz = (double)nodes[i].aabb_max[2] - nodes[i].aabb_min[2]; const double v = dx * dy + dy * dz + dz * dx; return v == v ? (v < 1e300 ? v : 1e300) : 0.0; };      // a total order even for infinite boxes
    for (size_t k = order.size(); k-- > 0; ) {                      // children before parents
        const int j = order[k], l = nodes[j].left_or_first;
        int slot[4], ns = 0;
        for (int c = l; c <= l + 1; c++) {
            if ((nodes[c].count & 0x3fffffff) > 0) slot[ns++] = c;
            else { slot[ns++] = nodes[c].left_or_first; slot[ns++] = nodes[c].left_or_first + 1; }
        }
        auto need_of = [&](const int * sl) { int nd = 0; for (int t = 0; t < ns; t++) nd = std::max(nd, (ns - 1 - t) + need[sl[t]]); return nd; };   // slot t is walked with ns-1-t siblings pending
        int nd = RTX_PK4_MAX_NEED + 1;
        if (order_mode != 0) { std::sort(slot, slot + ns, [&](int a, int b) { return area(a) > area(b); }); nd = need_of(slot); }
        if (nd > RTX_PK4_MAX_NEED - 4) { std::sort(slot, slot + ns, [&](int a, int b) { return need[a] < need[b]; }); nd = need_of(slot); }
        need[j] = nd;
        float4 * rec = &out[(size_t)4 * l];                         // 64 B per unit of `left`, 128 B per record (left is even)
        for (int s = 0; s < ns; s++) {
            const rtx_bvh_node & c = nodes[slot[s]];
            float f, cn; const int cnt = c.count & 0x3fffffff;
            memcpy(&f, &c.left_or_first, 4); memcpy(&cn, &cnt, 4);
            rec[2 * s]     = make_float4(c.aabb_min[0], c.aabb_min[1], c.aabb_max[0], c.aabb_max[1]);
            rec[2 * s + 1] = make_float4(c.aabb_min[2], c.aabb_max[2], f, cn);
        }
    }
    *stack_need = need[0];
    return true;
}

// The closest-hit per-lane walk's 4-wide records (rtx_packet.h, pk_lane_phase_closest).  BottomLevelBVH::trace (BottomLevelBVH.cpp:355-396)
// tests every node once, when it is popped, against the closest distance of that moment, and pushes an inner node's children untested, far
// child first.  With nested boxes (see build_nodes_pk4: AABB::intersect is monotone under nesting in floating point) a node that passes at
// its pop has a parent that passed at its own — earlier, hence against a distance at least as large — so the test of an inner child L of
// node X can be skipped: its children LL, LR meet, at THEIR pops, exactly the closest distances they meet in the reference (no leaf is visited
// between the pop of L and the pop of its near child), pass or fail as there, and reach the same leaves in the same order provided the four
// grandchildren are taken in the order the two binary levels would take them: X's near child's near child first.  The record of inner node
// j keeps the tree's shape for that: slots 0-1 = the left child's children (or the left child itself, a leaf, in slot 0), slots 2-3 the
// right child's; the near / far axes of the two children ride in slots 0 and 2.  Half the dependent fetches per walk.
//   record (128 B, at byte offset 64 * left-child index of j): 4 x { (min.x, min.y, max.x, max.y) (min.z, max.z, first, meta) },
//   meta = leaf count (< 16) | axis of the slot's PARENT (slots 0 and 2; 0: the parent is a leaf, i.e. the slot itself) << 26 | axis of the slot node << 30;
//   unused slots hold a point box, which never passes.  *stack_need = pending entries a lane's walk can have (bound over all visiting orders).
static bool build_nodes_pk4c(const rtx_bvh_node * nodes, int n, int tri_count, std::vector<float4> & out, int * stack_need) {
    if (n >= (1 << 24) || tri_count >= (1 << 24)) return false;
    auto inside = [&](int c, int p) {
        for (int a = 0; a < 3; a++) if (!(nodes[c].aabb_min[a] >= nodes[p].aabb_min[a] && nodes[c].aabb_max[a] <= nodes[p].aabb_max[a])) return false;
        return true;
    };
    std::vector<int> order, stack(1, 0);
    while (!stack.empty()) {
        const int i = stack.back(); stack.pop_back();
        const int cnt = nodes[i].count & 0x3fffffff, f = nodes[i].left_or_first;
        if (cnt > 0) { if (cnt >= 16) return false; continue; }      // work-list and stack entries pack (first, count) into one word
        if (f & 1) return false;                                   // records are addressed by left / 2
        if (!inside(f, i) || !inside(f + 1, i)) return false;      // nesting
        order.push_back(i); stack.push_back(f); stack.push_back(f + 1);
    }
    out.assign((size_t)4 * (n > 0 ? n : 1) + 8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<int> need((size_t)n, 0);
    for (size_t k = order.size(); k-- > 0; ) {                      // children before parents
        const int j = order[k], l = nodes[j].left_or_first;
        float4 * rec = &out[(size_t)4 * l];
        int nd = 0, ns = 0;
        for (int g = 0; g < 2; g++) {
            const rtx_bvh_node & c = nodes[l + g];
            const bool leaf = (c.count & 0x3fffffff) > 0;
            const int slots = leaf ? 1 : 2;
            for (int t = 0; t < slots; t++) {
                const int si = leaf ? l + g : c.left_or_first + t;
                const rtx_bvh_node & sn = nodes[si];
                const uint32_t meta = ((uint32_t)sn.count & 0x3fffffffu) | (t == 0 && !leaf ? (((uint32_t)c.count >> 30) << 26) : 0u) | (((uint32_t)sn.count >> 30) << 30);
                float f, m; memcpy(&f, &sn.left_or_first, 4); memcpy(&m, &meta, 4);
                rec[2 * (2 * g + t)]     = make_float4(sn.aabb_min[0], sn.aabb_min[1], sn.aabb_max[0], sn.aabb_max[1]);      // the packet kernels' slot layout: the slab test runs in packed fp32
                rec[2 * (2 * g + t) + 1] = make_float4(sn.aabb_min[2], sn.aabb_max[2], f, m);
                nd = std::max(nd, need[si]); ns++;
            }
        }
        need[j] = nd + ns - 1;                                      // any slot may come first, with the others pending
    }
    *stack_need = need[0];
    return true;
}

extern "C" int rtx_abi_version(void) { return RTX_ABI_VERSION; }

extern "C" const char * rtx_last_error(const rtx_ctx * ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int rtx_create(const rtx_config * config, rtx_ctx ** out_ctx) {
    if (!config || !out_ctx) return RTX_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    if (config->width <= 0 || config->height <= 0 || config->bounces < 0) return RTX_ERR_INVALID_ARG;
    if (config->bounces + 1 > RTX_MAX_LEVELS) return RTX_ERR_LIMIT;
    if (config->stack_size < 1 || config->stack_size > RTX_MAX_STACK) return RTX_ERR_LIMIT;
    if (config->traversal_strategy < 0 || config->traversal_strategy > 1 || config->texture_mode < 0 || config->texture_mode > 2 ||
        config->mip_filter < 0 || config->mip_filter > 2) return RTX_ERR_INVALID_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return RTX_ERR_NO_DEVICE;
    if (config->device < 0 || config->device >= n_dev) return RTX_ERR_NO_DEVICE;
    if (hipSetDevice(config->device) != hipSuccess) return RTX_ERR_NO_DEVICE;

    rtx_ctx * c = new rtx_ctx();
    c->cfg = *config;
    memset(&c->scene, 0, sizeof(c->scene));
    memset(&c->q, 0, sizeof(c->q));
    memset(&c->stats_acc, 0, sizeof(c->stats_acc));
    memset(&c->work_acc, 0, sizeof(c->work_acc));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, config->device) != hipSuccess) { delete c; return RTX_ERR_HIP; }
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return RTX_ERR_HIP; }
    c->stream = c->own_stream;
    // any_stream is created on first use (two-stream mode only): HIP multiplexes streams onto a few hardware queues, and an unused
    // stream per context would push other contexts' streams onto shared queues
    hipEventCreateWithFlags(&c->ev_shade0, hipEventDisableTiming); hipEventCreateWithFlags(&c->ev_shade_last, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_any_done, hipEventDisableTiming);

    // Texture::init(alpha = 2) EWA weight table, Texture.h:53-62 (host libm, like the reference; shipped as data)
    float ewa[RTX_EWA_LUT_SIZE];
    {
        const float alpha = 2.0f;
        const float denom = 1.0f / (float)(RTX_EWA_LUT_SIZE - 1);
        const float exp_neg_alpha = expf(-alpha);
        for (int i = 0; i < RTX_EWA_LUT_SIZE; i++) { float r2 = (float)i * denom; ewa[i] = expf(-alpha * r2) - exp_neg_alpha; }
    }
    int rc = upload(c, c->d_ewa, ewa, sizeof(ewa));
    // MaterialBuffer::init(): material 0 = black default, Material.h:52-60
    rtx_material m0; memset(&m0, 0, sizeof(m0)); m0.texture_id = -1; m0.index_of_refraction = 1.0f;
    if (!rc) rc = upload(c, c->d_materials, &m0, sizeof(m0));
    c->material_count = 1;
    // Window framebuffer, Window.cpp:42,76
    const size_t px = (size_t)config->width * config->height;
    if (!rc) rc = ensure(c, c->d_fb_rgb, px * 12);
    if (!rc) rc = ensure(c, c->d_fb_packed, px * 4);
    if (!rc) rc = ensure(c, c->d_counters, sizeof(DevCounters));
    if (!rc) rc = ensure(c, c->d_pk_heads, (size_t)2 * (RTX_MAX_LEVELS + 1) * RTX_PK_CLASSES * 32 * sizeof(uint32_t));
    float zero_sky[6] = { 0, 0, 0, 0, 0, 0 };
    if (!rc) rc = upload(c, c->d_sky, zero_sky, sizeof(zero_sky));
    c->sky_size = 1;
    if (rc) { rtx_destroy(c); return rc; }
    hipMemset(c->d_fb_rgb.p, 0, px * 12);
    hipMemset(c->d_fb_packed.p, 0, px * 4);

    // ---- knobs: parsed once, validated, clamped (see struct Knobs) ----
    Knobs & K = c->knobs;
    K.slot_budget = knob_int("RTX_SLOT_BUDGET", SLOT_BUDGET, 1024, 1ll << 31);
    K.item_bytes_max = knob_int("RTX_PK_ITEM_BYTES", 1ll << 30, 0, 1ll << 36);
    K.shade_grid = (int)knob_int("RTX_SHADE_GRID", 4, 1, 64);
    K.lane_from_level = (int)knob_int("RTX_LANE_FROM_LEVEL", -1, 0, 1 << 20);
    K.lane_from_level_any = (int)knob_int("RTX_LANE_FROM_LEVEL_ANY", -1, 0, 1 << 20);
    K.split_items = (int)knob_int("RTX_PK_SPLIT", 256, 0, 1 << 16);
    if (K.split_items > 0) K.split_items = (K.split_items + 63) & ~63;          // chunks are whole units of 64 items
    K.resolve_block = (int)knob_int("RTX_RESOLVE_BLOCK", 256, 64, 256);
    if (K.resolve_block != 64 && K.resolve_block != 128 && K.resolve_block != 256) K.resolve_block = 256;
    K.pk4_order = (int)knob_int("RTX_PK4_ORDER", 1, 0, 1);
    K.merge_any = knob_int("RTX_SERIAL_MERGE_ANY", 1, 0, 1) != 0;       // one-stream mode: ONE shadow-ray launch for all levels after the last shade (0: one per level)
    K.no_wide = knob_int("RTX_PK_WIDE", 1, 0, 1) == 0;                  // 0: binary shadow-ray walk for every mesh
    K.lpt = (int)knob_int("RTX_PK_LPT", -1, -1, 1);                     // level-0 closest-hit packets longest first (last frame's cost): 1 always, 0 never, -1: in the two-stream (one frame at a time) shape
    K.graph = knob_int("RTX_GRAPH", 0, 0, 1) != 0;                      // 1: identical rtx_render_tiles calls replay a captured hipGraph
    K.no_wide_closest = knob_int("RTX_PK_WIDE_CLOSEST", 1, 0, 1) == 0;  // 0: binary per-lane phase of the closest-hit walk for every mesh
    K.fuse_shade = knob_int("RTX_FUSE_SHADE", 0, 0, 1) != 0;      // measured: 1.51 vs 1.28 ms per cfg3 frame with three frames in flight: off
    K.fail_item_alloc = knob_int("RTX_DEBUG_FAIL_ITEM_ALLOC", 0, 0, 1) != 0;      // tests: the item buffer's allocation fails (a size no device has), the fallback kernel must take over
    int bpc = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_trace_fast<false>, RTX_TRACE_BLOCK, 0);
    c->trace_blocks_closest = c->n_cu * (bpc > 0 ? bpc : 2);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_trace_fast<true>, RTX_TRACE_BLOCK, 0);
    c->trace_blocks_any = c->n_cu * (bpc > 0 ? bpc : 2);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_trace<false, true>, RTX_TRACE_BLOCK, 0);
    c->trace_blocks_count = c->n_cu * (bpc > 0 ? bpc : 2);
    if (K.fuse_shade) hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_packet<false, false, false, true, false>, RTX_PK_BLOCK, 0);      // one grid size for every closest-hit packet launch of a context
    else              hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_packet<false, false>, RTX_PK_BLOCK, 0);
    c->pk_blocks_closest = (c->n_cu * (bpc > 0 ? bpc : 4) + 1) & ~1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_packet<true, false>, RTX_PK_BLOCK, 0);
    c->pk_blocks_any = (c->n_cu * (bpc > 0 ? bpc : 4) + 1) & ~1;
    {   // persistent-grid sizes of the packet kernels (A/B runs): a scale on the occupancy-derived size, or blocks per CU
        const double f = knob_real("RTX_PK_GRID_SCALE", 1.0, 0.05, 16.0);
        if (f != 1.0) { c->pk_blocks_closest = ((int)(c->pk_blocks_closest * f) + 1) & ~1; c->pk_blocks_any = ((int)(c->pk_blocks_any * f) + 1) & ~1; }
        const double fa = knob_real("RTX_PK_GRID_ANY", 0.0, 0.05, 64.0), fc = knob_real("RTX_PK_GRID_CLOSEST", 0.0, 0.05, 64.0);
        if (fa > 0.0) c->pk_blocks_any = ((int)(c->n_cu * fa) + 1) & ~1;
        if (fc > 0.0) c->pk_blocks_closest = ((int)(c->n_cu * fc) + 1) & ~1;
    }
    {   // the packet queues hand out packets k * 8 + head after a static first round of one packet per wave: whole rounds only, so waves % 8 == 0
        const int g = RTX_PK_CLASSES * RTX_WAVE / RTX_PK_BLOCK > 1 ? RTX_PK_CLASSES * RTX_WAVE / RTX_PK_BLOCK : 1;
        c->pk_blocks_any = (c->pk_blocks_any + g - 1) / g * g; c->pk_blocks_closest = (c->pk_blocks_closest + g - 1) / g * g;
    }
    // threads of the largest launch that indexes the per-thread spill / work-list regions with its global thread id
    long long max_threads = (long long)(c->trace_blocks_closest > c->trace_blocks_any ? c->trace_blocks_closest : c->trace_blocks_any) * RTX_TRACE_BLOCK;
    if ((long long)c->trace_blocks_count * RTX_TRACE_BLOCK > max_threads) max_threads = (long long)c->trace_blocks_count * RTX_TRACE_BLOCK;
    if ((long long)c->pk_blocks_any * RTX_PK_BLOCK > max_threads) max_threads = (long long)c->pk_blocks_any * RTX_PK_BLOCK;       // the packet kernels' per-lane phases spill there too
    if ((long long)c->pk_blocks_closest * RTX_PK_BLOCK > max_threads) max_threads = (long long)c->pk_blocks_closest * RTX_PK_BLOCK;
    // deferral thresholds of the hybrid walks are lane counts (0 .. 64), the growth a shift count: anything else would reach the asm walkers as garbage
    c->q.pk_defer_t0 = (int)knob_int("RTX_PK_DEFER", 8, 0, 64);
    c->q.pk_defer_leaf = (int)knob_int("RTX_PK_DEFER_LEAF", c->q.pk_defer_t0 / 2, 0, 64);
    c->q.pk_defer_t0_closest = (int)knob_int("RTX_PK_DEFER_CLOSEST", 64, 0, 64);
    c->q.pk_closest_asm = (int)knob_int("RTX_PK_CLOSEST_ASM", 1, 0, 1);
    c->q.pk_defer_t0_primary = (int)knob_int("RTX_PK_DEFER_PRIMARY", c->q.pk_closest_asm ? 4 : 64, 0, 64);
    c->q.pk_order = (int)knob_int("RTX_PK_ORDER", 1, 0, 1);
    c->q.prof_level = (int)knob_int("RTX_LANE_PROF_LEVEL", 2, 0, 100 + RTX_MAX_LEVELS);      // -DRTX_LANE_PROF builds only (tools/lane_prof.py)
    c->q.pk_defer_grow = (int)knob_int("RTX_PK_GROW", 3, 0, 31);
    c->q.spill_threads = (int)max_threads;
    rc = ensure(c, c->d_spill, (size_t)c->q.spill_threads * (RTX_MAX_STACK - RTX_LDS_STACK) * 4 * 3);   // three regions, see k_trace_fast
    if (!rc) rc = ensure(c, c->d_pk_fifo, (size_t)c->q.spill_threads * RTX_PK_FIFO * 8);        // (reference, key) per work-list entry
    { int ib = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&ib, k_items, RTX_ITEM_BLOCK, 0); c->item_blocks = c->n_cu * (ib > 0 ? ib : 4);
      const double f = knob_real("RTX_ITEM_GRID", 0.0, 0.05, 64.0); if (f > 0.0) c->item_blocks = (int)(c->n_cu * f) > 0 ? (int)(c->n_cu * f) : 1; }
    c->q.pk_items = nullptr; c->q.pk_item_count = nullptr; c->q.pk_item_cap = 0;      // sized per render call (render_tiles_impl)
    if (rc) { rtx_destroy(c); return rc; }
    *out_ctx = c;
    return RTX_OK;
}

extern "C" int rtx_destroy(rtx_ctx * c) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto & v : c->blas_allocs) for (void * p : v) hipFree(p);
    for (void * p : c->tex_allocs) if (p) hipFree(p);
    DevBuf * bufs[] = { &c->d_blas, &c->d_materials, &c->d_textures, &c->d_sky, &c->d_ewa, &c->d_frame, &c->d_counters, &c->d_spill, &c->d_fb_rgb, &c->d_fb_packed, &c->d_display, &c->d_gamma, &c->d_pk_heads, &c->d_pk_fifo, &c->d_pk_items, &c->d_pk_item_count, &c->d_stats_partial, &c->d_dbg_s0, &c->d_dbg_s1, &c->d_pk_cost, &c->d_pk_order };
    for (DevBuf * b : bufs) if (b->p) hipFree(b->p);
    for (DevBuf & b : c->qb) if (b.p) hipFree(b.p);
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    for (DevCounters * h : c->pending_host) hipHostFree(h);
    for (auto & st : c->stage) { if (st.host) hipHostFree(st.host); if (st.done) hipEventDestroy(st.done); }
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->order_stream) { hipStreamSynchronize(c->order_stream); hipStreamDestroy(c->order_stream); hipEventDestroy(c->ev_cost); hipEventDestroy(c->ev_order); }
    if (c->any_stream) { hipStreamSynchronize(c->any_stream); hipStreamDestroy(c->any_stream); }
    if (c->ev_shade0) hipEventDestroy(c->ev_shade0);
    if (c->ev_shade_last) hipEventDestroy(c->ev_shade_last);
    if (c->ev_any_done) hipEventDestroy(c->ev_any_done);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return RTX_OK;
}

// Walks the nodes a traversal can reach from the root (abandoned child pairs and index 1 hold garbage in reference-built arrays and are
// not looked at): every leaf range inside the primitive array, every child pair inside the node array, and no node reachable twice —
// a cycle would keep a GPU traversal going for ever.
static bool validate_tree(const rtx_bvh_node * nodes, int node_count, int64_t primitive_count, int * max_inner_depth = nullptr, int * max_leaf_count = nullptr) {
    if (node_count < 1) return false;
    std::vector<unsigned char> seen((size_t)node_count, 0);
    std::vector<std::pair<int, int>> stack(1, std::make_pair(0, 0));       // node, depth
    int deepest = -1, biggest = 0;
    while (!stack.empty()) {
        const int i = stack.back().first, d = stack.back().second; stack.pop_back();
        if (seen[i]) return false;
        seen[i] = 1;
        const int cnt = nodes[i].count & 0x3fffffff, f = nodes[i].left_or_first;
        if (cnt > 0) { if (f < 0 || (int64_t)f + cnt > primitive_count) return false; if (cnt > biggest) biggest = cnt; }
        else { if (f < 0 || f + 1 >= node_count) return false; if (d > deepest) deepest = d; stack.push_back(std::make_pair(f, d + 1)); stack.push_back(std::make_pair(f + 1, d + 1)); }
    }
    if (max_inner_depth) *max_inner_depth = deepest;
    if (max_leaf_count) *max_leaf_count = biggest;
    return true;
}


extern "C" int rtx_upload_blas(rtx_ctx * c, int32_t blas_id, const rtx_bvh_node * nodes, int32_t node_count,
                               const rtx_triangle_hot * tri_hot, const rtx_triangle_cold * tri_cold,
                               int32_t triangle_count, int32_t material_offset) {
    if (!c || blas_id < 0 || blas_id >= (1 << 20) || !nodes || node_count <= 0 || triangle_count < 0 || (triangle_count > 0 && (!tri_hot || !tri_cold)))
        return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    // every leaf must address triangles inside the arrays, every inner node children inside the node array
    int inner_depth = -1, leaf_max = 0;
    if (!validate_tree(nodes, node_count, triangle_count, &inner_depth, &leaf_max)) return RTX_ERR_INVALID_ARG;
    if ((size_t)blas_id >= c->h_blas.size()) { c->h_blas.resize(blas_id + 1); c->blas_allocs.resize(blas_id + 1); for (auto & b : c->h_blas) if (!b.nodes) memset(&b, 0, sizeof(b)); }
    for (void * p : c->blas_allocs[blas_id]) hipFree(p);
    c->blas_allocs[blas_id].clear();

    std::vector<float4> nd; convert_nodes(nodes, node_count, nd);
    std::vector<float4> th((size_t)RTX_TRI_STRIDE * (triangle_count > 0 ? triangle_count : 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (int i = 0; i < triangle_count; i++) {
        th[RTX_TRI_STRIDE * (size_t)i]     = make_float4(tri_hot[i].position_0[0], tri_hot[i].position_0[1], tri_hot[i].position_0[2], 0.0f);
        th[RTX_TRI_STRIDE * (size_t)i + 1] = make_float4(tri_hot[i].position_edge_1[0], tri_hot[i].position_edge_1[1], tri_hot[i].position_edge_1[2], 0.0f);
        th[RTX_TRI_STRIDE * (size_t)i + 2] = make_float4(tri_hot[i].position_edge_2[0], tri_hot[i].position_edge_2[1], tri_hot[i].position_edge_2[2], 0.0f);
    }
    std::vector<float4> ndp; convert_nodes_pk(nodes, node_count, ndp);
    std::vector<float4> nd4; int need4 = 0;
    const bool wide = !c->knobs.no_wide && build_nodes_pk4(nodes, node_count, triangle_count, nd4, &need4, c->knobs.pk4_order) && need4 <= RTX_PK4_MAX_NEED;
    std::vector<float4> nd4c; int need4c = 0;
    const bool wide_closest = !c->knobs.no_wide_closest && build_nodes_pk4c(nodes, node_count, triangle_count, nd4c, &need4c) && need4c <= RTX_MAX_STACK - 2;
    // the distinct box-plane coordinates per axis, ascending: a ray with a zero direction component can only produce a NaN in a slab test
    // (0 * inf) if its origin lies exactly on one of them (pk_nan_possible, rtx_packet.h)
    std::vector<float> planes[3];
    for (int a = 0; a < 3; a++) {
        planes[a].reserve((size_t)node_count * 2);
        for (int i = 0; i < node_count; i++) { planes[a].push_back(nodes[i].aabb_min[a]); planes[a].push_back(nodes[i].aabb_max[a]); }
        planes[a].erase(std::remove_if(planes[a].begin(), planes[a].end(), [](float v) { return v != v; }), planes[a].end());
        std::sort(planes[a].begin(), planes[a].end());
        planes[a].erase(std::unique(planes[a].begin(), planes[a].end()), planes[a].end());      // +0 and -0 compare equal: one entry
        if (planes[a].empty()) planes[a].push_back(INFINITY);
    }
    void * dn = nullptr, * dt = nullptr, * dc = nullptr, * dnp = nullptr, * dn4 = nullptr, * dn4c = nullptr, * dpl[3] = { nullptr, nullptr, nullptr };
    for (int a = 0; a < 3; a++) { HIP_OK(c, hipMalloc(&dpl[a], planes[a].size() * 4)); c->blas_allocs[blas_id].push_back(dpl[a]); HIP_OK(c, hipMemcpy(dpl[a], planes[a].data(), planes[a].size() * 4, hipMemcpyHostToDevice)); }
    if (wide_closest) { HIP_OK(c, hipMalloc(&dn4c, nd4c.size() * 16)); c->blas_allocs[blas_id].push_back(dn4c); HIP_OK(c, hipMemcpy(dn4c, nd4c.data(), nd4c.size() * 16, hipMemcpyHostToDevice)); }
    if (wide) { HIP_OK(c, hipMalloc(&dn4, nd4.size() * 16)); c->blas_allocs[blas_id].push_back(dn4); HIP_OK(c, hipMemcpy(dn4, nd4.data(), nd4.size() * 16, hipMemcpyHostToDevice)); }
    HIP_OK(c, hipMalloc(&dn, nd.size() * 16)); c->blas_allocs[blas_id].push_back(dn);
    HIP_OK(c, hipMalloc(&dnp, ndp.size() * 16)); c->blas_allocs[blas_id].push_back(dnp);
    HIP_OK(c, hipMemcpy(dnp, ndp.data(), ndp.size() * 16, hipMemcpyHostToDevice));
    HIP_OK(c, hipMalloc(&dt, th.size() * 16)); c->blas_allocs[blas_id].push_back(dt);
    HIP_OK(c, hipMalloc(&dc, (size_t)(triangle_count > 0 ? triangle_count : 1) * sizeof(rtx_triangle_cold))); c->blas_allocs[blas_id].push_back(dc);
    HIP_OK(c, hipMemcpy(dn, nd.data(), nd.size() * 16, hipMemcpyHostToDevice));
    HIP_OK(c, hipMemcpy(dt, th.data(), th.size() * 16, hipMemcpyHostToDevice));
    if (triangle_count) HIP_OK(c, hipMemcpy(dc, tri_cold, (size_t)triangle_count * sizeof(rtx_triangle_cold), hipMemcpyHostToDevice));
    if (c->blas_max_local_material.size() <= (size_t)blas_id) c->blas_max_local_material.resize(blas_id + 1, -1);
    int max_local = -1;
    for (int i = 0; i < triangle_count; i++) { if (tri_cold[i].material_id < 0) return RTX_ERR_INVALID_ARG; if (tri_cold[i].material_id > max_local) max_local = tri_cold[i].material_id; }
    c->blas_max_local_material[blas_id] = max_local; c->refs_dirty = true;
    if (c->blas_inner_depth.size() <= (size_t)blas_id) { c->blas_inner_depth.resize(blas_id + 1, -1); c->blas_packet_ok.resize(blas_id + 1, 1); }
    c->blas_inner_depth[blas_id] = inner_depth;
    // the packet kernels pack stack entries and address nodes / triangles with 32-bit byte offsets (rtx_packet.h)
    c->blas_packet_ok[blas_id] = (leaf_max < 65536 && node_count < (1 << 26) && triangle_count < (1 << 25)) ? 1 : 0;
    DevBlas & B = c->h_blas[blas_id];
    B.nodes = (const float4 *)dn; B.pk_nodes = (const float4 *)dnp; B.tri_hot = (const float4 *)dt; B.tri_cold = (const rtx_triangle_cold *)dc;
    B.pk4_nodes = (const float4 *)dn4; B.pk4c_nodes = (const float4 *)dn4c;
    for (int a = 0; a < 3; a++) { B.planes[a] = (const float *)dpl[a]; B.plane_count[a] = (int32_t)planes[a].size(); }
    B.node_count = node_count; B.tri_count = triangle_count; B.material_offset = material_offset; B.pk4_need = wide ? need4 : -1; B.pk4c_need = wide_closest ? need4c : -1;
    return upload(c, c->d_blas, c->h_blas.data(), c->h_blas.size() * sizeof(DevBlas));
}

extern "C" int rtx_upload_materials(rtx_ctx * c, const rtx_material * materials, int32_t count) {
    if (!c || !materials || count <= 0) return RTX_ERR_INVALID_ARG;
    if (count > RTX_MAX_MATERIALS) return RTX_ERR_LIMIT;       // "Max Material limit reached!" Material.h:33-37
    hipSetDevice(c->cfg.device);
    c->material_count = count;
    c->h_materials.assign(materials, materials + count); c->refs_dirty = true;
    return upload(c, c->d_materials, materials, (size_t)count * sizeof(rtx_material));
}

extern "C" int rtx_upload_texture(rtx_ctx * c, int32_t texture_id, const rtx_texture_desc * desc, const float * texels_rgb, int64_t texel_count) {
    if (!c || texture_id < 0 || texture_id >= 4096 || !desc || !texels_rgb) return RTX_ERR_INVALID_ARG;
    if (desc->width <= 0 || desc->height <= 0 || desc->mip_levels < 1 || desc->mip_levels > RTX_MAX_MIP_LEVELS) return RTX_ERR_INVALID_ARG;
    for (int l = 0; l < desc->mip_levels; l++) {
        const int64_t lw = desc->width >> l, lh = desc->height >> l;
        if (lw < 1 || lh < 1 || desc->mip_offsets[l] < 0 || desc->mip_offsets[l] + lw * lh > texel_count) return RTX_ERR_INVALID_ARG;
    }
    hipSetDevice(c->cfg.device);
    if ((size_t)texture_id >= c->h_tex.size()) { size_t old = c->h_tex.size(); c->h_tex.resize(texture_id + 1); for (size_t i = old; i < c->h_tex.size(); i++) memset(&c->h_tex[i], 0, sizeof(DevTexture)); }
    void * d = nullptr;
    HIP_OK(c, hipMalloc(&d, (size_t)texel_count * 16));
    if (c->h_tex[texture_id].texels) {            // re-upload of an id: the old texel array is released (after the frames that may still read it)
        HIP_OK(c, hipStreamSynchronize(c->stream));
        for (void *& p : c->tex_allocs) if (p == (void *)c->h_tex[texture_id].texels) { hipFree(p); p = nullptr; }
    }
    c->tex_allocs.push_back(d);
    {   // float3 texels of the ABI -> one float4 per texel on the device
        std::vector<float4> padded((size_t)texel_count);
        for (int64_t i = 0; i < texel_count; i++) padded[(size_t)i] = make_float4(texels_rgb[3 * i], texels_rgb[3 * i + 1], texels_rgb[3 * i + 2], 0.0f);
        HIP_OK(c, hipMemcpy(d, padded.data(), (size_t)texel_count * 16, hipMemcpyHostToDevice));
    }
    c->h_tex[texture_id].desc = *desc;
    c->h_tex[texture_id].texels = (const float4 *)d;
    c->refs_dirty = true;
    return upload(c, c->d_textures, c->h_tex.data(), c->h_tex.size() * sizeof(DevTexture));
}

extern "C" int rtx_upload_sky(rtx_ctx * c, const float * texels_rgb, int32_t size) {
    if (!c || !texels_rgb || size <= 0 || size > 16384) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    // Sky::sample clamps the texel index to size*size INCLUSIVE (Sky.cpp:45): one zero texel of padding
    std::vector<float> padded((size_t)size * size * 3 + 3, 0.0f);
    memcpy(padded.data(), texels_rgb, (size_t)size * size * 12);
    c->sky_size = size;
    return upload(c, c->d_sky, padded.data(), padded.size() * 4);
}

extern "C" int rtx_set_frame(rtx_ctx * c, const rtx_frame * f) {
    if (!c || !f) return RTX_ERR_INVALID_ARG;
    if (f->tlas_node_count < 0 || f->instance_count < 0 || f->sphere_count < 0 || f->plane_count < 0 || f->tlas_index_count < 0 ||
        f->point_light_count < 0 || f->spot_light_count < 0 || f->directional_light_count < 0) return RTX_ERR_INVALID_ARG;
    if ((f->tlas_node_count && !f->tlas_nodes) || (f->instance_count && !f->instances) || (f->tlas_index_count && !f->tlas_indices) ||
        (f->sphere_count && !f->spheres) || (f->plane_count && !f->planes) || (f->point_light_count && !f->point_lights) ||
        (f->spot_light_count && !f->spot_lights) || (f->directional_light_count && !f->directional_lights)) return RTX_ERR_INVALID_ARG;
    if (f->instance_count >= (1 << 28) || f->sphere_count >= (1 << 28) || f->plane_count >= (1 << 28)) return RTX_ERR_LIMIT;
    hipSetDevice(c->cfg.device);
    // validate the TLAS against the instance table and the instances against the uploaded BLAS set
    int tlas_depth = -1;
    if (f->tlas_node_count > 0 && !validate_tree(f->tlas_nodes, f->tlas_node_count, f->tlas_index_count, &tlas_depth)) return RTX_ERR_INVALID_ARG;
    for (int i = 0; i < f->tlas_index_count; i++) if (f->tlas_indices[i] < 0 || f->tlas_indices[i] >= f->instance_count) return RTX_ERR_INVALID_ARG;
    for (int i = 0; i < f->instance_count; i++) {
        const int b = f->instances[i].blas_id;
        if (b < 0 || (size_t)b >= c->h_blas.size() || !c->h_blas[b].nodes) return RTX_ERR_STATE;
    }
    // Pack the frame into one block: [TLAS nodes, lane layout][TLAS nodes, packet layout][indices][instances][spheres][planes][lights],
    // every part 256-byte aligned.  The block is staged in pinned memory and copied by ONE hipMemcpyAsync on the context's stream: it is
    // ordered after the frames already queued there (which still read the previous contents) and before the next render call.
    std::vector<float4> nd, ndp; convert_nodes(f->tlas_nodes, f->tlas_node_count, nd); convert_nodes_pk(f->tlas_nodes, f->tlas_node_count, ndp);
    const void * src[9] = { nd.data(), ndp.data(), f->tlas_indices, f->instances, f->spheres, f->planes, f->point_lights, f->spot_lights, f->directional_lights };
    const size_t len[9] = { nd.size() * 16, ndp.size() * 16, (size_t)f->tlas_index_count * 4, (size_t)f->instance_count * sizeof(rtx_instance),
                            (size_t)f->sphere_count * sizeof(rtx_sphere), (size_t)f->plane_count * sizeof(rtx_plane), (size_t)f->point_light_count * sizeof(rtx_point_light),
                            (size_t)f->spot_light_count * sizeof(rtx_spot_light), (size_t)f->directional_light_count * sizeof(rtx_directional_light) };
    size_t off[9], total = 0;
    for (int k = 0; k < 9; k++) { off[k] = total; total += (len[k] + 255) & ~(size_t)255; if (len[k] == 0) total += 256; }
    int rc = RTX_OK;
    if (total > c->d_frame.cap) {                           // growth (first frame, or more instances than ever before): the one case that waits
        HIP_OK(c, hipStreamSynchronize(c->stream));
        rc = ensure(c, c->d_frame, total + total / 2);
        if (rc) return rc;
    }
    rtx_ctx::FrameStage & st = c->stage[c->stage_next]; c->stage_next = (c->stage_next + 1) % 3;
    if (st.pending) { HIP_OK(c, hipEventSynchronize(st.done)); st.pending = false; }      // two frames back: long finished in steady state
    if (total > st.cap) {
        if (st.host) hipHostFree(st.host);
        st.host = nullptr; st.cap = 0;
        HIP_OK(c, hipHostMalloc(&st.host, total + total / 2, hipHostMallocDefault)); st.cap = total + total / 2;
    }
    if (!st.done) HIP_OK(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    for (int k = 0; k < 9; k++) if (len[k]) memcpy((char *)st.host + off[k], src[k], len[k]);
    HIP_OK(c, hipMemcpyAsync(c->d_frame.p, st.host, total, hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipEventRecord(st.done, c->stream)); st.pending = true;
    char * const fb = (char *)c->d_frame.p;
    if (rc) return rc;

    DevScene & s = c->scene;
    s.width = c->cfg.width; s.height = c->cfg.height; s.bounces = c->cfg.bounces; s.stack_size = c->cfg.stack_size;
    s.traversal_strategy = c->cfg.traversal_strategy; s.texture_mode = c->cfg.texture_mode; s.mip_filter = c->cfg.mip_filter;
    s.heatmap = c->cfg.heatmap ? 1 : 0;
    s.diff_enabled = c->cfg.texture_mode == RTX_TEXTURE_MIPMAP;            // RAY_DIFFERENTIALS_ENABLED, Config.h:46
    s.max_anisotropy = c->cfg.max_anisotropy;
    s.tile_count_x = (c->cfg.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;   // Window.cpp:11
    memcpy(s.cam_pos, f->camera.position, 12); memcpy(s.cam_tl, f->camera.rotated_top_left_corner, 12);
    memcpy(s.cam_x, f->camera.rotated_x_axis, 12); memcpy(s.cam_y, f->camera.rotated_y_axis, 12);
    memcpy(s.ambient, f->ambient, 12);
    s.blas = (const DevBlas *)c->d_blas.p; s.blas_count = (int)c->h_blas.size();
    s.materials = (const rtx_material *)c->d_materials.p; s.material_count = c->material_count;
    s.textures = (const DevTexture *)c->d_textures.p; s.texture_count = (int)c->h_tex.size();
    s.sky = (const float *)c->d_sky.p; s.sky_size = c->sky_size;
    s.ewa_table = (const float *)c->d_ewa.p;
    s.tlas_nodes = (const float4 *)(fb + off[0]); s.tlas_node_count = f->tlas_node_count; s.pk_tlas_nodes = (const float4 *)(fb + off[1]);
    s.tlas_indices = (const int32_t *)(fb + off[2]); s.tlas_index_count = f->tlas_index_count;
    s.instances = (const rtx_instance *)(fb + off[3]); s.instance_count = f->instance_count;
    s.spheres = (const rtx_sphere *)(fb + off[4]); s.sphere_count = f->sphere_count;
    s.planes = (const rtx_plane *)(fb + off[5]); s.plane_count = f->plane_count;
    s.point_lights = (const rtx_point_light *)(fb + off[6]); s.point_light_count = f->point_light_count;
    s.spot_lights = (const rtx_spot_light *)(fb + off[7]); s.spot_light_count = f->spot_light_count;
    s.dir_lights = (const rtx_directional_light *)(fb + off[8]); s.dir_light_count = f->directional_light_count;
    s.light_count = f->point_light_count + f->spot_light_count + f->directional_light_count;
    c->frame_primitive_materials.clear();
    for (int i = 0; i < f->sphere_count; i++) c->frame_primitive_materials.push_back(f->spheres[i].material_id);
    for (int i = 0; i < f->plane_count; i++) c->frame_primitive_materials.push_back(f->planes[i].material_id);
    c->tlas_inner_depth = tlas_depth;
    c->frame_set = true; c->refs_dirty = true;
    return RTX_OK;
}

// Every index a kernel dereferences without a bounds check of its own — material ids of primitives and triangles, texture ids of
// materials — is verified here, once after any upload / rtx_set_frame: a wrong id must become a status code, not a GPU memory fault.
static int validate_references(rtx_ctx * c) {
    if (!c->refs_dirty) return RTX_OK;
    const int nm = c->material_count;
    for (int id : c->frame_primitive_materials) if (id < 0 || id >= nm) { c->err = "sphere / plane material id outside the uploaded material table"; return RTX_ERR_STATE; }
    for (size_t b = 0; b < c->h_blas.size(); b++) {
        if (!c->h_blas[b].nodes) continue;
        const int hi = b < c->blas_max_local_material.size() ? c->blas_max_local_material[b] : -1;
        if (c->h_blas[b].material_offset < 0 || (hi >= 0 && (long long)c->h_blas[b].material_offset + hi >= nm)) { c->err = "BLAS material_offset + triangle material id outside the uploaded material table"; return RTX_ERR_STATE; }
    }
    // BVH_TRAVERSAL_STACK_SIZE (Config.h:25).  The reference's per-BVH stack holds one pending sibling per ancestor entered through its near
    // child plus the two children pushed while a node is expanded (BottomLevelBVH.cpp:381-387, TopLevelBVH.cpp:76-82) and is not checked:
    // a tree with an inner node at depth d overflows a stack of fewer than d + 2 entries for some ray.  Such a tree is refused here,
    // for every ray alike.  The per-lane kernels' stacks (64 entries per BVH) can then never overflow; the packet kernels share one 64-entry
    // stack between TLAS and BLAS, and render_tiles_impl hands scenes that could exceed it to the per-lane kernels.
    for (size_t b = 0; b < c->h_blas.size(); b++)
        if (c->h_blas[b].nodes && b < c->blas_inner_depth.size() && c->blas_inner_depth[b] + 2 > c->cfg.stack_size) {
            c->err = "BVH deeper than rtx_config.stack_size allows (BVH_TRAVERSAL_STACK_SIZE, Config.h:25): an inner node at depth " + std::to_string(c->blas_inner_depth[b]) + " needs " + std::to_string(c->blas_inner_depth[b] + 2) + " stack entries";
            return RTX_ERR_LIMIT;
        }
    if (c->tlas_inner_depth + 2 > c->cfg.stack_size) { c->err = "TLAS deeper than rtx_config.stack_size allows (BVH_TRAVERSAL_STACK_SIZE, Config.h:25)"; return RTX_ERR_LIMIT; }
    for (size_t i = 0; i < c->h_materials.size(); i++) {
        const int t = c->h_materials[i].texture_id;
        if (t >= 0 && ((size_t)t >= c->h_tex.size() || !c->h_tex[t].texels)) { c->err = "material refers to a texture id that was never uploaded"; return RTX_ERR_STATE; }
    }
    c->refs_dirty = false;
    return RTX_OK;
}

// ---- queue memory ------------------------------------------------------------------------------------
static int plan_batch(rtx_ctx * c, int tiles, int & out_tiles) {
    // worst case: every hit spawns two children -> level d holds P * 2^d rays
    const int levels = c->cfg.bounces + 1;
    const long long per_tile = 1024ll * ((1ll << levels) - 1);
    const long long budget = c->knobs.slot_budget;          // RTX_SLOT_BUDGET: tests force multi-batch frames with a small budget
    long long t = budget / per_tile;
    if (t < 1) t = 1;
    out_tiles = (int)(t < tiles ? t : tiles);
    return RTX_OK;
}

static int alloc_queues(rtx_ctx * c, int batch_tiles) {
    const int levels = c->cfg.bounces + 1;
    const long long P = 1024ll * batch_tiles;
    long long total = 0;
    // fused shading hands out the slots of levels >= 1 in wave-private chunks (rtx_packet.h RTX_PK_CHUNK): every wave of the launch above may leave one partly used
    const long long chunk_slack = c->knobs.fuse_shade ? (long long)c->pk_blocks_closest * (RTX_PK_BLOCK / RTX_WAVE) * RTX_PK_CHUNK : 0;
    for (int d = 0; d <= RTX_MAX_LEVELS; d++) {
        long long cap = d < levels ? (P << d) + (d > 0 ? chunk_slack : 0) : 0;
        c->q.level_base[d] = (int32_t)total; c->q.level_cap[d] = (int32_t)cap;
        total += cap;
    }
    if (total >= (1ll << 31)) { c->err = "batch too large"; return RTX_ERR_LIMIT; }
    const int nL = c->scene.light_count;
    long long stotal = 0;
    for (int d = 0; d <= RTX_MAX_LEVELS; d++) { c->q.shadow_base[d] = (int32_t)stotal; stotal += (long long)c->q.level_cap[d] * nL; }
    if (stotal >= (1ll << 31)) { c->err = "shadow queue too large"; return RTX_ERR_LIMIT; }
    const size_t slots = (size_t)total, sslots = (size_t)(stotal > 0 ? stotal : 1);
    int rc = 0, k = 0;
    // rays of level 0 are never stored, but the arrays are indexed by global slot for simplicity of addressing
    void ** f4[] = { (void **)&c->q.r0, (void **)&c->q.r1, (void **)&c->q.r2, (void **)&c->q.r3, (void **)&c->q.r4, (void **)&c->q.h0,
                     (void **)&c->q.n0, (void **)&c->q.n1, (void **)&c->q.n2, (void **)&c->q.c0, (void **)&c->q.c1,
                     (void **)&c->q.sp, (void **)&c->q.sn };
    for (void ** p : f4) { if (!rc) rc = ensure(c, c->qb[k], slots * 16); *p = c->qb[k].p; k++; }
    if (!rc) rc = ensure(c, c->qb[k], slots * 4); c->q.h1 = (int32_t *)c->qb[k].p; k++;
    c->q.n3 = nullptr; c->q.s0 = c->q.s1 = nullptr; c->q.shadow_explicit = 0; c->q.cull = 0;      // explicit shadow rays: rtx_debug_occluded allocates them (debug_explicit_rays)
    if (!rc) rc = ensure(c, c->qb[k], sslots * 4); c->q.socc = (uint32_t *)c->qb[k].p; k++;
    c->q.spill = (int32_t *)c->d_spill.p;
    c->q.counters = (DevCounters *)c->d_counters.p;
    c->q.pk_heads = (uint32_t *)c->d_pk_heads.p;
    c->q.pk_fifo = (int32_t *)c->d_pk_fifo.p;
    c->q.fb_rgb = (float *)(c->ext_rgb ? c->ext_rgb : c->d_fb_rgb.p);
    c->q.fb_packed = (uint32_t *)(c->ext_packed ? c->ext_packed : c->d_fb_packed.p);
    return rc;
}

__global__ void k_begin_batch(DevCounters * ctr, uint32_t * pk_heads, uint32_t primary_slots, uint32_t closest_threads, uint32_t any_threads,
                              uint32_t pk_waves_closest, uint32_t pk_waves_any, int first_batch) {
    // the first batch of a render call zeroes the whole counter block (stats of this call, WorkerThread.cpp:120) — done here rather than
    // with hipMemsetAsync, whose blit serialises the streams of different contexts against each other; every batch then resets the
    // queue state: the dynamic-fetch heads start behind the statically assigned first round (one ray per resident thread)
    const int i = threadIdx.x;
    if (first_batch) {
        uint32_t * w = reinterpret_cast<uint32_t *>(ctr);
        for (int k = i; k < (int)(sizeof(DevCounters) / 4); k += blockDim.x) w[k] = 0u;
        __syncthreads();
    }
    if (i <= RTX_MAX_LEVELS) { ctr->ray_count[i] = (i == 0) ? primary_slots : 0u; ctr->item_max[i] = 0u; ctr->fetch_closest[i] = closest_threads; ctr->fetch_any[i] = any_threads; }
    // packet-queue heads (rtx_packet.h): head c of a launch hands out packets k * 8 + c; the static first round (wave w takes packet w)
    // has already covered k < waves / 8 of every head
    for (int k = i; k < 2 * (RTX_MAX_LEVELS + 1) * RTX_PK_CLASSES; k += blockDim.x)
        pk_heads[k * 32] = (k < (RTX_MAX_LEVELS + 1) * RTX_PK_CLASSES ? pk_waves_closest : pk_waves_any) / RTX_PK_CLASSES;
}

static hipEvent_t next_event(rtx_ctx * c) {
    if (c->event_next == c->event_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->event_pool.push_back(e); }
    return c->event_pool[c->event_next++];
}

// Packets by descending cost: a counting sort over 256 linear cost classes in ONE workgroup (32 400 packets at 1080p: ~10 us, on a side stream).
// The order inside a class is whatever the atomics make it — it is a schedule, not a result.
__global__ __launch_bounds__(1024) void k_packet_order(const uint32_t * __restrict__ cost, uint32_t * __restrict__ order, const uint32_t n) {
    __shared__ uint32_t hist[256], base[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    auto cls = [](uint32_t c) { const uint32_t k = c >> 7; return 255u - (k < 255u ? k : 255u); };      // class 0 = the most expensive
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[cls(cost[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int k = 0; k < 256; k++) { base[k] = acc; acc += hist[k]; } }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) order[atomicAdd(&base[cls(cost[i])], 1u)] = i;
}

template <typename F>
static void launch_timed(rtx_ctx * c, const char * name, hipStream_t stream, F && launch) {
    if (c->timing) {
        KernelTime kt; kt.name = name; kt.a = next_event(c); kt.b = next_event(c);
        hipEventRecord(kt.a, stream);                 // events on the stream the kernel is launched on
        launch();
        hipEventRecord(kt.b, stream);
        c->times.push_back(kt);
    } else launch();
}

static int render_tiles_impl(rtx_ctx * c, int32_t first_tile, int32_t tile_stride, int32_t tile_count, uint32_t flags, uint32_t * tile_major);

extern "C" int rtx_render_tiles(rtx_ctx * c, int32_t first_tile, int32_t tile_stride, int32_t tile_count, uint32_t flags) {
    return render_tiles_impl(c, first_tile, tile_stride, tile_count, flags, nullptr);
}

// tile_major != nullptr: the packed level-0 pixels of the i-th rendered tile go to tile_major[i * 1024 ...] (slot order) instead of the framebuffer
static int render_tiles_impl(rtx_ctx * c, int32_t first_tile, int32_t tile_stride, int32_t tile_count, uint32_t flags, uint32_t * tile_major) {
    if (!c || first_tile < 0 || tile_stride < 1 || tile_count < 0) return RTX_ERR_INVALID_ARG;
    if (!c->frame_set) { c->err = "rtx_render_tiles before rtx_set_frame"; return RTX_ERR_STATE; }
    if (int bad = validate_references(c)) return bad;
    const int tcx = (c->cfg.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE, tcy = (c->cfg.height + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    if (tile_count > 0 && first_tile + (int64_t)(tile_count - 1) * tile_stride >= (int64_t)tcx * tcy) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    const bool count_work = (flags & RTX_RENDER_COUNT_WORK) != 0;
    const bool simple = (flags & RTX_RENDER_SIMPLE_TRACE) != 0;
    const bool cull = (flags & RTX_RENDER_CULL_DEAD_SHADOW_RAYS) != 0 && !simple && !count_work;
    bool lane = (flags & RTX_RENDER_LANE_TRACE) != 0;
    for (size_t b = 0; b < c->blas_packet_ok.size(); b++) if (c->h_blas[b].nodes && !c->blas_packet_ok[b]) lane = true;      // limits of the packet kernels' packed entries
    bool pk_closest = (flags & RTX_RENDER_PACKET_CLOSEST) != 0;            // closest-hit packets walk shared subtrees together (default: lanes turn private at once)
    {   // The packet kernels keep ONE 64-entry stack per wave (RTX_PK_STACK) for the TLAS part and the BLAS part of a walk together, where the
        // reference has a stack per BVH (BVH_TRAVERSAL_STACK_SIZE each).  Both depths are known here, so the choice is made on the host
        // and pk_push's overflow path is never taken:  TLAS part = one pending far sibling per level + the iterator entry of the leaf being
        // visited (closest-hit rays: + one parked sign-split entry per level); BLAS part = the 4-wide records' bound (pk4_need), or one far
        // sibling per level of the binary walk; the shared closest-hit walk parks sign-split entries there too.  Scenes beyond the bound
        // (e.g. a chain-shaped TLAS of 60 instances) are traced by the per-lane kernels, whose stacks are per BVH like the reference's.
        const int dt = c->tlas_inner_depth < 0 ? 0 : c->tlas_inner_depth + 1;
        int blas_any = 0, blas_shared = 0;
        for (size_t b = 0; b < c->h_blas.size(); b++) {
            if (!c->h_blas[b].nodes) continue;
            const int depth = b < c->blas_inner_depth.size() ? c->blas_inner_depth[b] + 2 : 2;
            const int any = c->h_blas[b].pk4_nodes ? c->h_blas[b].pk4_need : depth;
            if (any > blas_any) blas_any = any;
            if (2 * depth > blas_shared) blas_shared = 2 * depth;
        }
        if (dt + 1 + blas_any > RTX_PK_STACK || 2 * dt + 1 > RTX_PK_STACK) lane = true;
        if (2 * dt + 1 + blas_shared > RTX_PK_STACK) pk_closest = false;
    }
    const bool pstat = (flags & RTX_RENDER_PACKET_STATS) != 0 && !simple && !count_work && !lane;
    c->serial = (flags & RTX_RENDER_SERIAL) != 0;

    // stats of this call are reset by the first k_begin_batch (WorkerThread.cpp:120 zeroes them per frame)
    if (tile_count == 0) { hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, c->stream, (DevCounters *)c->d_counters.p, (uint32_t *)c->d_pk_heads.p, 0u, 0u, 0u, 8u, 8u, 1); c->stats_pending = true; return RTX_OK; }

    int batch_tiles = 0;
    plan_batch(c, tile_count, batch_tiles);
    int rc = alloc_queues(c, batch_tiles);
    if (rc) return rc;
    // Levels that can hold rays: a hit spawns a reflection / refraction ray only where its material's Ks / Kt is not all zero (Raytracer.cpp:204-213,
    // rtx_shade.h reflection_mask / refraction_mask).  With no such material uploaded the levels >= 1 are provably empty, and their
    // launches — three per level, each a floor of 7-9 us — are not queued (BASELINE configs[1]: diffuse Monkey.obj with NUMBER_OF_BOUNCES 3).
    bool can_spawn = false;
    for (const rtx_material & m : c->h_materials)
        for (int a = 0; a < 3; a++) if (m.reflection[a] != 0.0f || m.transmittance[a] != 0.0f) can_spawn = true;      // != is true for NaN too: conservative
    const int levels = can_spawn ? c->cfg.bounces + 1 : 1;
    const int stream_blocks = c->n_cu * 8;          // k_resolve (256 threads)
    const int shade_blocks = c->n_cu * c->knobs.shade_grid;      // k_shade (RTX_SHADE_BLOCK = 256 threads, 3 resident blocks per CU at 168 VGPRs)
    const int pk_waves_closest = c->pk_blocks_closest * (RTX_PK_BLOCK / RTX_WAVE);
    const int stats_stride = shade_blocks > pk_waves_closest ? shade_blocks : pk_waves_closest;      // per level: one entry per k_shade workgroup, or per wave of the fused packet kernel
    rc = ensure(c, c->d_stats_partial, (size_t)(RTX_MAX_LEVELS + 1) * stats_stride * 4 * sizeof(uint32_t));
    if (rc) return rc;

    // Item chunks of the split shadow-ray walk (rtx_packet.h, k_items): one chunk per wave of the packet launch, sized ONCE per call for its
    // largest batch — ≈0.3 items per shadow ray in the cfg3 frame, the fullest chunk 3x the average, so 3.5x the expected average, in
    // whole units of 64 items.  The buffer only ever grows, never beyond knobs.item_bytes_max (1 GiB; a chunk that fills up merely makes the
    // packet keep its nodes, which the kernel supports), and only here, before anything of this call is queued.  If the allocation
    // fails the pending HIP error is cleared and the call continues with the non-split kernel: a correctly queued frame must not
    // report RTX_ERR_HIP.  The split walk needs 4-wide records for every mesh.
    c->q.pk_items = nullptr; c->q.pk_item_count = nullptr; c->q.pk_item_cap = 0;
    if (c->knobs.split_items > 0 && c->scene.light_count > 0) {
        bool all_wide = true;
        for (size_t b = 0; b < c->h_blas.size(); b++) if (c->h_blas[b].nodes && !c->h_blas[b].pk4_nodes) all_wide = false;
        if (all_wide) {
            const size_t chunks = (size_t)c->pk_blocks_any * (RTX_PK_BLOCK / RTX_WAVE);
            const double expect = 0.45 * 1024.0 * (double)batch_tiles * c->scene.light_count / (double)chunks;
            long long cap = c->knobs.split_items;
            if (cap < (long long)(3.5 * expect)) cap = ((long long)(3.5 * expect) + 63) & ~63ll;
            const long long cap_max = (c->knobs.item_bytes_max / (long long)(chunks * 48)) & ~63ll;
            if (cap > cap_max) cap = cap_max;
            if (cap > (1 << 16)) cap = 1 << 16;
            if (cap >= 64) {
                if (cap <= c->item_cap_alloc && c->d_pk_items.p && !c->knobs.fail_item_alloc) cap = c->item_cap_alloc;      // the buffer is there already
                else {
                    c->item_cap_alloc = 0;
                    const size_t bytes = c->knobs.fail_item_alloc ? ((size_t)1 << 46) : chunks * (size_t)cap * 48;
                    if (ensure(c, c->d_pk_items, bytes) == 0 && ensure(c, c->d_pk_item_count, chunks * 4) == 0) c->item_cap_alloc = (int)cap;
                    else { (void)hipGetLastError(); c->err.clear(); cap = 0; }       // fall back to the in-kernel per-lane phase
                }
                if (cap >= 64) { c->q.pk_items = (uint4 *)c->d_pk_items.p; c->q.pk_item_count = (uint32_t *)c->d_pk_item_count.p; c->q.pk_item_cap = (int)cap; }
            }
        }
    }

    // hipGraph replay: the kernel arguments of a call are functions of (scene, queues, tile range, flags, knobs); while those bytes stay the same the
    // captured graph IS the call.  Only for single-batch calls without per-kernel timing.
    // RTX_PK_LPT: the level-0 closest-hit launch takes its packets longest first, by what the same packets cost in the previous call of this context
    // over the same tiles (costs written by the launch itself, sorted by k_packet_order on a side stream while the rest of the frame runs).
    // Measured (cfg3): the launch 309 -> 223 us alone, one frame at a time 1.48 -> 1.41 ms; with three frames in flight the tail it removes was
    // being filled by the other frames' kernels anyway (1.090 -> 1.098 ms), and a launch with one packet per wave has nothing to reorder: so by default only
    // in the two-stream shape and with at least two packets per wave.
    const uint32_t lpt_n = (uint32_t)tile_count * 16u;
    const bool lpt_want = c->knobs.lpt > 0 || (c->knobs.lpt < 0 && !c->serial && lpt_n >= 2u * (uint32_t)pk_waves_closest);
    const bool lpt = lpt_want && !c->knobs.graph && tile_count <= batch_tiles && !count_work && !simple && !lane && !pstat && !c->scene.heatmap && !c->knobs.fuse_shade;
    c->q.pk_cost = nullptr; c->q.pk_lpt_order = nullptr;
    if (lpt) {
        int lrc = ensure(c, c->d_pk_cost, (size_t)lpt_n * 4); if (!lrc) lrc = ensure(c, c->d_pk_order, (size_t)lpt_n * 4);
        if (lrc) return lrc;
        if (!c->order_stream) { HIP_OK(c, hipStreamCreateWithFlags(&c->order_stream, hipStreamNonBlocking)); hipEventCreateWithFlags(&c->ev_cost, hipEventDisableTiming); hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming); }
        const int32_t key[4] = { first_tile, tile_stride, tile_count, levels };
        const bool use_order = c->lpt_valid && memcmp(key, c->lpt_key, sizeof(key)) == 0;
        memcpy(c->lpt_key, key, sizeof(key));
        c->q.pk_cost = (uint32_t *)c->d_pk_cost.p;
        if (use_order) { c->q.pk_lpt_order = (const uint32_t *)c->d_pk_order.p; hipStreamWaitEvent(c->stream, c->ev_order, 0); }
        else if (c->lpt_valid) hipStreamWaitEvent(c->stream, c->ev_order, 0);      // the previous call's sort still reads the cost buffer
    } else c->lpt_valid = false;
    bool capturing = false;
    if (c->knobs.graph && !c->timing && tile_count <= batch_tiles) {
        std::vector<unsigned char> key(sizeof(DevScene) + sizeof(DevQueues) + 5 * sizeof(int32_t) + sizeof(void *));
        unsigned char * kp = key.data();
        memcpy(kp, &c->scene, sizeof(DevScene)); kp += sizeof(DevScene); memcpy(kp, &c->q, sizeof(DevQueues)); kp += sizeof(DevQueues);
        const int32_t kv[5] = { first_tile, tile_stride, tile_count, (int32_t)flags, levels }; memcpy(kp, kv, sizeof(kv)); kp += sizeof(kv);
        memcpy(kp, &tile_major, sizeof(void *));
        if (c->graph_exec && key == c->graph_key) {
            HIP_OK(c, hipGraphLaunch(c->graph_exec, c->stream));
            c->stats_pending = true;
            return RTX_OK;
        }
        if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
        if (c->graph_warm == key) {           // second identical call: capture (the first one ran eagerly: lazy allocations, stream creation)
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) { capturing = true; c->graph_key = key; }
            else hipGetLastError();
        } else c->graph_warm = key;
    }
    for (int done = 0; done < tile_count; done += batch_tiles) {
        const int n = (tile_count - done) < batch_tiles ? (tile_count - done) : batch_tiles;
        DevQueues q = c->q;
        if (pk_closest) q.pk_defer_t0_closest = q.pk_defer_t0_primary = q.pk_defer_t0;
        q.first_tile = first_tile + done * tile_stride; q.tile_stride = tile_stride; q.tile_count = n; q.primary_slots = n * 1024;
        q.tm_packed = tile_major; q.tm_base = done * 1024;
        q.cull = cull ? 1 : 0;
        q.stats_partial = (uint32_t *)c->d_stats_partial.p; q.stats_stride = stats_stride;
        for (int d = 0; d <= RTX_MAX_LEVELS; d++) q.stats_n[d] = 0;
        const DevScene sc = c->scene;
        const bool plain = count_work || simple || sc.heatmap;
        hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, c->stream, q.counters, q.pk_heads, (uint32_t)q.primary_slots,
                           (uint32_t)((plain ? c->trace_blocks_count : c->trace_blocks_closest) * RTX_TRACE_BLOCK),
                           (uint32_t)((plain ? c->trace_blocks_count : c->trace_blocks_any) * RTX_TRACE_BLOCK),
                           (uint32_t)(c->pk_blocks_closest * (RTX_PK_BLOCK / RTX_WAVE)), (uint32_t)(c->pk_blocks_any * (RTX_PK_BLOCK / RTX_WAVE)), done == 0 ? 1 : 0);
        // Main stream: closest(d) -> shade(d) for d = 0..D.  Shadow rays depend only on shade, so in the default (fast)
        // configuration they run on a second stream: any(level 0) starts after shade(0) and overlaps the deeper levels'
        // closest/shade kernels; the shadow rays of levels 1..D are traced by ONE more launch after shade(D).  This removes
        // three of the per-launch tails (a persistent trace launch has a ~0.17 ms floor set by its slowest rays).
        const bool overlap = !count_work && !simple && sc.light_count > 0 && !c->serial;
        if (overlap && !c->any_stream) HIP_OK(c, hipStreamCreateWithFlags(&c->any_stream, hipStreamNonBlocking));
        const bool merged = !overlap && !count_work && !simple && c->knobs.merge_any;
        if (sc.heatmap) {
            // BVH_VISUALIZE_HEATMAP: bounce() returns right after the primary ray's trace (Raytracer.cpp:97-102), so a frame is one
            // closest-hit pass in reference pop order (the plain kernel counts the steps) and one colouring pass
            launch_timed(c, "k_trace_closest", c->stream, [&] {
                if (count_work) hipLaunchKernelGGL((k_trace<false, true>),  dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, 0);
                else            hipLaunchKernelGGL((k_trace<false, false>), dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, 0); });
            launch_timed(c, "k_heatmap", c->stream, [&] { hipLaunchKernelGGL(k_heatmap, dim3(stream_blocks), dim3(256), 0, c->stream, sc, q); });
            continue;
        }
        // traversal kernels: packet walk (production), per-lane pair fetch (RTX_RENDER_LANE_TRACE), plain pop-and-test (SIMPLE / COUNT_WORK)
        // Which kernel for which level: a packet's walk of the TLAS costs the UNION of its rays' instances, each entered with its own
        // transform and per-lane phase, so for the incoherent rays of the deeper levels of a multi-instance scene the per-lane kernels
        // (refill, per-lane TLAS walk) win by 2.5-4x (cfg5: closest-hit levels 2 / 3 0.43 / 0.45 -> 0.16 / 0.12 ms, shadow rays 0.41 / 0.35 ->
        // 0.19 / 0.14 ms) while the packet kernels win everywhere else (tools/perlevel3.py).  Both produce the same bits.
        // With hundreds of instances the closest-hit reflection rays of level 1 already prefer the per-lane kernel (tools/many_instances.py:
        // 144 instances 0.52 vs 0.40 ms, 576 instances 0.70 vs 0.39 ms), the shadow rays of level 1 do not (0.48 vs 0.8 ms).
        int lane_from = c->knobs.lane_from_level >= 0 ? c->knobs.lane_from_level : (sc.instance_count > 1 ? 2 : RTX_MAX_LEVELS + 1);
        const int lane_from_closest = c->knobs.lane_from_level >= 0 ? c->knobs.lane_from_level : (sc.instance_count > 64 ? 1 : lane_from);
        // With the split walk a shadow-ray packet that enters an instance with few lanes hands the whole visit over as items, and ONE packet
        // launch for all levels beats a packet launch + a per-lane launch at every instance count measured (tools/any_rule.sh: 16 / 144 / 576
        // instances 1.54 / 2.24 / 3.01 vs 1.66 / 2.41 / 3.05 ms per frame, cfg5 2.32 vs 2.42): shadow rays then stay with the packets
        if (c->knobs.lane_from_level < 0 && c->knobs.split_items > 0 && sc.light_count > 0) { bool all_wide = true; for (size_t b = 0; b < c->h_blas.size(); b++) if (c->h_blas[b].nodes && !c->h_blas[b].pk4_nodes) all_wide = false; if (all_wide) lane_from = RTX_MAX_LEVELS + 1; }
        if (c->knobs.lane_from_level_any >= 0) lane_from = c->knobs.lane_from_level_any;      // shadow rays only (A/B runs)
        // Which levels shade their own hits inside the closest-hit packet kernel (k_packet<.., FUSE>, rtx_packet.h) and which get a k_shade
        // launch (every other closest-hit kernel: per-lane, plain, instrumented).  Decided before anything is launched: k_resolve is told
        // how many partial tallies each level's shading pass leaves (q.stats_n).
        auto level_is_fused = [&](int level) { return c->knobs.fuse_shade && !count_work && !simple && !pstat && !lane && level < lane_from_closest; };
        // Grids of the streaming kernels (k_shade, k_resolve: grid-stride loops, any grid is correct) follow the batch: level d of a batch of P
        // primary slots is given room for P / 2^d rays — the dispatcher spends ~16 ns per workgroup, which is most of a small launch's
        // time (a 1/8 tile shard's k_shade launches of levels 1-3: 18 us each with 1 024 workgroups for 74 k / 6 k / 1 k rays).
        auto level_blocks = [&](int level, int full) { const long long want = (((long long)q.primary_slots >> level) + 255) / 256; return (int)std::max(64ll, std::min((long long)full, want)); };
        for (int level = 0; level < levels; level++) q.stats_n[level] = level_is_fused(level) ? pk_waves_closest : level_blocks(level, shade_blocks);
        auto launch_closest = [&](int level) {
            launch_timed(c, level_is_fused(level) ? "k_trace_closest_shade" : "k_trace_closest", c->stream, [&] {
                if (count_work)     hipLaunchKernelGGL((k_trace<false, true>),   dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, level);
                else if (simple)    hipLaunchKernelGGL((k_trace<false, false>),  dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, level);
                else if (lane || (!pstat && level >= lane_from_closest)) hipLaunchKernelGGL((k_trace_fast<false>),    dim3(c->trace_blocks_closest), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, level, level);
                else if (pstat)     hipLaunchKernelGGL((k_packet<false, true>),  dim3(c->pk_blocks_closest), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, level, level);
                else if (level_is_fused(level) && cull) hipLaunchKernelGGL((k_packet<false, false, false, true, true>),  dim3(c->pk_blocks_closest), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, level, level);
                else if (level_is_fused(level))         hipLaunchKernelGGL((k_packet<false, false, false, true, false>), dim3(c->pk_blocks_closest), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, level, level);
                else                hipLaunchKernelGGL((k_packet<false, false>), dim3(c->pk_blocks_closest), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, level, level);
            });
        };
        // split shadow-ray walk (RTX_PK_SPLIT): q.pk_items / q.pk_item_cap were sized for this call's largest batch before the loop
        const bool split = q.pk_items != nullptr;
        auto launch_items = [&](hipStream_t st, int lo) {
            if (!split) return;
            launch_timed(c, "k_trace_items", st, [&] { hipLaunchKernelGGL(k_items, dim3(c->item_blocks), dim3(RTX_ITEM_BLOCK), 0, st, sc, q, c->pk_blocks_any * (RTX_PK_BLOCK / RTX_WAVE), lo); });
        };
        auto launch_any = [&](hipStream_t st, int lo, int hi) {
            if (!count_work && !simple && !lane && !pstat && hi >= lane_from) {      // levels lo .. lane_from-1 by packets, the rest per lane
                if (lo < lane_from) { launch_timed(c, "k_trace_any", st, [&] { if (split) hipLaunchKernelGGL((k_packet<true, false, true>), dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, st, sc, q, lo, lane_from - 1);
                                                                                 else       hipLaunchKernelGGL((k_packet<true, false>), dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, st, sc, q, lo, lane_from - 1); }); launch_items(st, lo); }
                const int l0 = lo > lane_from ? lo : lane_from;
                launch_timed(c, "k_trace_any", st, [&] { hipLaunchKernelGGL((k_trace_fast<true>), dim3(c->trace_blocks_any), dim3(RTX_TRACE_BLOCK), 0, st, sc, q, l0, hi); });
                return;
            }
            launch_timed(c, "k_trace_any", st, [&] {
                if (count_work)     hipLaunchKernelGGL((k_trace<true, true>),   dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, st, sc, q, lo);
                else if (simple)    hipLaunchKernelGGL((k_trace<true, false>),  dim3(c->trace_blocks_count), dim3(RTX_TRACE_BLOCK), 0, st, sc, q, lo);
                else if (lane)      hipLaunchKernelGGL((k_trace_fast<true>),    dim3(c->trace_blocks_any), dim3(RTX_TRACE_BLOCK), 0, st, sc, q, lo, hi);
                else if (pstat)     hipLaunchKernelGGL((k_packet<true, true>),  dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, st, sc, q, lo, hi);
                else if (split)     hipLaunchKernelGGL((k_packet<true, false, true>), dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, st, sc, q, lo, hi);
                else                hipLaunchKernelGGL((k_packet<true, false>), dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, st, sc, q, lo, hi);
            });
            if (!count_work && !simple && !lane && !pstat) launch_items(st, lo);
        };
        for (int level = 0; level < levels; level++) {
            launch_closest(level);
            if (lpt && level == 0) {
                hipEventRecord(c->ev_cost, c->stream); hipStreamWaitEvent(c->order_stream, c->ev_cost, 0);
                hipLaunchKernelGGL(k_packet_order, dim3(1), dim3(1024), 0, c->order_stream, (const uint32_t *)c->d_pk_cost.p, (uint32_t *)c->d_pk_order.p, lpt_n);
                hipEventRecord(c->ev_order, c->order_stream);
                c->lpt_valid = true;
            }
            if (!level_is_fused(level)) launch_timed(c, "k_shade", c->stream, [&] {
                const int g = level_blocks(level, shade_blocks);
                if (count_work) hipLaunchKernelGGL((k_shade<true, false>),  dim3(g), dim3(RTX_SHADE_BLOCK), 0, c->stream, sc, q, level);
                else if (cull)  hipLaunchKernelGGL((k_shade<false, true>),  dim3(g), dim3(RTX_SHADE_BLOCK), 0, c->stream, sc, q, level);
                else            hipLaunchKernelGGL((k_shade<false, false>), dim3(g), dim3(RTX_SHADE_BLOCK), 0, c->stream, sc, q, level);
            });
            if (sc.light_count == 0) continue;
            if (overlap) {
                if (level == 0) {
                    hipEventRecord(c->ev_shade0, c->stream);
                    hipStreamWaitEvent(c->any_stream, c->ev_shade0, 0);
                    launch_any(c->any_stream, 0, 0);
                }
                if (level == levels - 1 && levels > 1) {
                    hipEventRecord(c->ev_shade_last, c->stream);
                    hipStreamWaitEvent(c->any_stream, c->ev_shade_last, 0);
                    launch_any(c->any_stream, 1, levels - 1);
                }
            } else if (merged) {
                // shadow rays only feed k_resolve: all levels' shadow rays are traced by ONE launch after the last shade
                if (level == levels - 1) launch_any(c->stream, 0, levels - 1);
            } else launch_any(c->stream, level, level);
        }
        if (overlap) { hipEventRecord(c->ev_any_done, c->any_stream); hipStreamWaitEvent(c->stream, c->ev_any_done, 0); }
        for (int level = levels - 1; level >= 0; level--)
            launch_timed(c, "k_resolve", c->stream, [&] { hipLaunchKernelGGL(k_resolve, dim3(level_blocks(level, stream_blocks) * (256 / c->knobs.resolve_block)), dim3(c->knobs.resolve_block), 0, c->stream, sc, q, level); });
    }
    if (capturing) {
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(c->stream, &g) != hipSuccess || !g) { hipGetLastError(); c->graph_key.clear(); c->err = "hipGraph capture failed"; return RTX_ERR_HIP; }
        hipError_t e = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) { hipGetLastError(); c->graph_exec = nullptr; c->graph_key.clear(); c->err = "hipGraphInstantiate failed"; return RTX_ERR_HIP; }
        HIP_OK(c, hipGraphLaunch(c->graph_exec, c->stream));
    }
    HIP_OK(c, hipGetLastError());
    c->stats_pending = true;
    return RTX_OK;
}

extern "C" int rtx_synchronize(rtx_ctx * c) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

extern "C" int rtx_get_stats(rtx_ctx * c, rtx_stats * stats, rtx_work_counters * work) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    if (c->stats_pending) {
        DevCounters h;
        HIP_OK(c, hipMemcpyAsync(&h, c->d_counters.p, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(c, hipStreamSynchronize(c->stream));
        c->stats_acc.num_primary_rays = h.stats[0]; c->stats_acc.num_shadow_rays = h.stats[1];
        c->stats_acc.num_reflection_rays = h.stats[2]; c->stats_acc.num_refraction_rays = h.stats[3];
        memset(&c->work_acc, 0, sizeof(c->work_acc));
        memcpy(&c->work_acc, h.work, sizeof(uint64_t) * 20);
        c->err_flags_acc = h.error_flags;
        c->stats_pending = false;
    }
    if (stats) *stats = c->stats_acc;
    if (work)  *work  = c->work_acc;
    if (c->err_flags_acc & ERR_STACK_OVERFLOW) { c->err = "BVH traversal stack overflow (BVH_TRAVERSAL_STACK_SIZE, Config.h:25)"; return RTX_ERR_LIMIT; }
    if (c->err_flags_acc & ERR_QUEUE_OVERFLOW) { c->err = "ray queue overflow"; return RTX_ERR_LIMIT; }
    return RTX_OK;
}

extern "C" int rtx_read_framebuffer(rtx_ctx * c, float * rgb_f32, uint32_t * packed_u32) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    const size_t px = (size_t)c->cfg.width * c->cfg.height;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (rgb_f32)    HIP_OK(c, hipMemcpy(rgb_f32, c->ext_rgb ? c->ext_rgb : c->d_fb_rgb.p, px * 12, hipMemcpyDeviceToHost));
    if (packed_u32) HIP_OK(c, hipMemcpy(packed_u32, c->ext_packed ? c->ext_packed : c->d_fb_packed.p, px * 4, hipMemcpyDeviceToHost));
    return RTX_OK;
}

// Window::draw_quad (Window.cpp:87-95): gamma 1/2.2 + optional FXAA over the packed frame, see rtx_present.h (parity unpinned vs GL)
extern "C" int rtx_present(rtx_ctx * c, int32_t enable_fxaa, uint32_t * display_u32, void ** display_dev) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    const size_t px = (size_t)c->cfg.width * c->cfg.height;
    int rc = ensure(c, c->d_display, px * 4);
    if (rc) return rc;
    if (!c->d_gamma.p) {
        rc = ensure(c, c->d_gamma, 256 * sizeof(float));
        if (rc) return rc;
        float lut[256];
        for (int i = 0; i < 256; i++) lut[i] = powf((float)i / 255.0f, 1.0f / 2.2f);         // pow(texel, vec3(1 / 2.2)), once per 8-bit value (host libm)
        HIP_OK(c, hipMemcpy(c->d_gamma.p, lut, sizeof(lut), hipMemcpyHostToDevice));
    }
    PresentArgs a;
    a.packed = (const uint32_t *)(c->ext_packed ? c->ext_packed : c->d_fb_packed.p);
    a.gamma_lut = (const float *)c->d_gamma.p; a.display = (uint32_t *)c->d_display.p;
    a.width = c->cfg.width; a.height = c->cfg.height; a.fxaa = enable_fxaa ? 1 : 0;
    hipLaunchKernelGGL(k_present, dim3((c->cfg.width + 15) / 16, (c->cfg.height + 15) / 16), dim3(256), 0, c->stream, a);   // after the frame, same stream
    HIP_OK(c, hipGetLastError());
    if (display_u32) {
        HIP_OK(c, hipStreamSynchronize(c->stream));
        HIP_OK(c, hipMemcpy(display_u32, c->d_display.p, px * 4, hipMemcpyDeviceToHost));
    }
    if (display_dev) *display_dev = c->d_display.p;
    return RTX_OK;
}

extern "C" int rtx_framebuffer_device_ptrs(rtx_ctx * c, void ** rgb_f32_dev, void ** packed_u32_dev) {
    if (!c) return RTX_ERR_INVALID_ARG;
    if (rgb_f32_dev) *rgb_f32_dev = c->ext_rgb ? c->ext_rgb : c->d_fb_rgb.p;
    if (packed_u32_dev) *packed_u32_dev = c->ext_packed ? c->ext_packed : c->d_fb_packed.p;
    return RTX_OK;
}

extern "C" int rtx_bind_framebuffer(rtx_ctx * c, void * rgb_f32_dev, void * packed_u32_dev) {
    if (!c || ((rgb_f32_dev == nullptr) != (packed_u32_dev == nullptr))) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    c->ext_rgb = rgb_f32_dev; c->ext_packed = packed_u32_dev;
    return RTX_OK;
}

extern "C" int rtx_set_stream(rtx_ctx * c, void * hip_stream) {
    if (!c) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RTX_OK;
}

extern "C" int rtx_enable_kernel_timing(rtx_ctx * c, int32_t enable) {
    if (!c) return RTX_ERR_INVALID_ARG;
    c->timing = enable != 0;
    c->times.clear(); c->event_next = 0;      // launches are recorded from now on, across render calls
    return RTX_OK;
}

extern "C" int rtx_last_kernel_times(rtx_ctx * c, const char ** names, float * ms, int32_t capacity, int32_t * count) {
    if (!c || !count) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    *count = (int32_t)c->times.size();
    for (int i = 0; i < (int)c->times.size() && i < capacity; i++) {
        float t = 0.0f;
        hipEventElapsedTime(&t, c->times[i].a, c->times[i].b);
        if (names) names[i] = c->times[i].name;
        if (ms) ms[i] = t;
    }
    return RTX_OK;
}

// ---- unit-level entry points (parity tests of single reference functions on the device) --------------
__global__ void k_debug_libm(int fn, const float * a, const float * b, float * out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = rtx_acosf(a[i]); break;
        case 1: r = rtx_atan2f(a[i], b[i]); break;
        case 2: r = rtx_expf(a[i]); break;
        case 3: r = rtx_log2f(a[i]); break;
        case 4: r = rtx_atanf(a[i]); break;
        case 5: r = (float)f2i_rn_x86(a[i]); break;
        default: r = 1.0f / sqrtf(a[i]); break;
    }
    out[i] = r;
}

extern "C" int rtx_debug_libm(rtx_ctx * c, int32_t fn, const float * a, const float * b, float * out, int32_t n) {
    if (!c || !a || !out || n <= 0) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    float * da = nullptr, * db = nullptr, * dout = nullptr;
    HIP_OK(c, hipMalloc(&da, (size_t)n * 4)); HIP_OK(c, hipMalloc(&db, (size_t)n * 4)); HIP_OK(c, hipMalloc(&dout, (size_t)n * 4));
    HIP_OK(c, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(c, hipMemcpy(db, b ? b : a, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_libm, dim3((n + 255) / 256), dim3(256), 0, c->stream, fn, da, db, dout, n);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dout);
    return RTX_OK;
}

__global__ void k_debug_texture(DevScene sc, int tex, const float * in6, float * out3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TexCtx tc; tc.t = &sc.textures[tex]; tc.fetches = 0;
    const float * p = in6 + 6 * (size_t)i;
    const v3 r = texture_sample(tc, sc, p[0], p[1], p[2], p[3], p[4], p[5]);
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
}

// Texture::sample (Texture.h:33-49) at n (s, t, ds_dx, ds_dy, dt_dx, dt_dy) inputs with the context's texture_mode / mip_filter
extern "C" int rtx_debug_texture_sample(rtx_ctx * c, int32_t texture_id, const float * in6, float * out3, int32_t n) {
    if (!c || !in6 || !out3 || n <= 0 || texture_id < 0 || (size_t)texture_id >= c->h_tex.size() || !c->h_tex[texture_id].texels) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    DevScene sc = c->scene;
    sc.texture_mode = c->cfg.texture_mode; sc.mip_filter = c->cfg.mip_filter; sc.max_anisotropy = c->cfg.max_anisotropy;
    sc.textures = (const DevTexture *)c->d_textures.p; sc.ewa_table = (const float *)c->d_ewa.p;
    float * din = nullptr, * dout = nullptr;
    HIP_OK(c, hipMalloc(&din, (size_t)n * 24)); HIP_OK(c, hipMalloc(&dout, (size_t)n * 12));
    HIP_OK(c, hipMemcpy(din, in6, (size_t)n * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_texture, dim3((n + 255) / 256), dim3(256), 0, c->stream, sc, texture_id, din, dout, n);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(out3, dout, (size_t)n * 12, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout);
    return RTX_OK;
}

__global__ void k_debug_sky(const float * sky, int size, const float * dirs, float * out3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 r = sky_sample(sky, size, V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
}

// which shadow-ray walk an uploaded mesh gets: the stack need of its 4-wide records, or -1 for the binary walk
extern "C" int rtx_debug_blas_wide(rtx_ctx * c, int32_t blas_id, int32_t * stack_need) {
    if (!c || !stack_need || blas_id < 0 || (size_t)blas_id >= c->h_blas.size() || !c->h_blas[blas_id].nodes) return RTX_ERR_INVALID_ARG;
    *stack_need = c->h_blas[blas_id].pk4_nodes ? c->h_blas[blas_id].pk4_need : -1;
    return RTX_OK;
}

// which per-lane closest-hit walk an uploaded mesh gets: the stack need of its ordered 4-wide records, or -1 for the binary walk
extern "C" int rtx_debug_blas_wide_closest(rtx_ctx * c, int32_t blas_id, int32_t * stack_need) {
    if (!c || !stack_need || blas_id < 0 || (size_t)blas_id >= c->h_blas.size() || !c->h_blas[blas_id].nodes) return RTX_ERR_INVALID_ARG;
    *stack_need = c->h_blas[blas_id].pk4c_nodes ? c->h_blas[blas_id].pk4c_need : -1;
    return RTX_OK;
}

// Sky::sample (Sky.cpp:28-68) at n directions
extern "C" int rtx_debug_sky_sample(rtx_ctx * c, const float * dirs3, float * out3, int32_t n) {
    if (!c || !dirs3 || !out3 || n <= 0) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    float * din = nullptr, * dout = nullptr;
    HIP_OK(c, hipMalloc(&din, (size_t)n * 12)); HIP_OK(c, hipMalloc(&dout, (size_t)n * 12));
    HIP_OK(c, hipMemcpy(din, dirs3, (size_t)n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_sky, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float *)c->d_sky.p, c->sky_size, din, dout, n);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(out3, dout, (size_t)n * 12, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout);
    return RTX_OK;
}


// ---- GPU groups: screen tiles sharded over the GPUs of one node, RCCL gather of the packed tiles to rank 0 -----------
// RCCL is bound at run time (dlopen): librtx_hip.so has no link-time dependency on it, and a process that already carries an RCCL
// (e.g. PyTorch's) gets that same copy.
#include <dlfcn.h>
typedef struct { char internal[128]; } rtx_nccl_id;
typedef void * rtx_nccl_comm;
struct RcclApi {
    void * lib = nullptr;
    int (*GetUniqueId)(rtx_nccl_id *) = nullptr;
    int (*CommInitRank)(rtx_nccl_comm *, int, rtx_nccl_id, int) = nullptr;
    int (*CommInitAll)(rtx_nccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(rtx_nccl_comm) = nullptr;
    int (*Gather)(const void *, void *, size_t, int, int, rtx_nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
};
static RcclApi * rccl() {
    static RcclApi api; static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    const char * names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char * n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; }
    if (!api.lib) return nullptr;
    bool ok = true;
    auto sym = [&](const char * n) { void * p = dlsym(api.lib, n); if (!p) ok = false; return p; };
    api.GetUniqueId = (int (*)(rtx_nccl_id *))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(rtx_nccl_comm *, int, rtx_nccl_id, int))sym("ncclCommInitRank");
    api.CommInitAll = (int (*)(rtx_nccl_comm *, int, const int *))sym("ncclCommInitAll");
    api.CommDestroy = (int (*)(rtx_nccl_comm))sym("ncclCommDestroy");
    api.Gather = (int (*)(const void *, void *, size_t, int, int, rtx_nccl_comm, hipStream_t))sym("ncclGather");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.GetErrorString = (const char * (*)(int))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.lib); api.lib = nullptr; return nullptr; }
    return &api;
}

// One communicator per rank, shared by every context of that rank that joins it (rtx_group_attach: further frames in flight on the same GPU).
// With more than one member the gathers are issued on the hub's own exchange stream, in call order: every rank issues them in the same
// order whatever order its frames finish rendering in, and no two collectives of a rank are ever in flight at once.
struct GroupHub { rtx_nccl_comm comm = nullptr; hipStream_t xs = nullptr; int refs = 1; int device = 0; };
struct rtx_group {
    rtx_ctx * ctx = nullptr;
    int rank = 0, world = 1, tiles_total = 0, tiles_per_rank = 0, own_tiles = 0;
    GroupHub * hub = nullptr;
    hipEvent_t ev_rendered = nullptr, ev_gathered = nullptr;
    DevBuf send, recv;           // tile-major packed pixels: this rank's tiles; on rank 0 every rank's
};

static int group_setup(rtx_group * g) {
    rtx_ctx * c = g->ctx;
    const int tcx = (c->cfg.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE, tcy = (c->cfg.height + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    g->tiles_total = tcx * tcy;
    g->tiles_per_rank = group_tiles_per_rank(g->tiles_total, g->world);      // padded: every rank sends the same count
    g->own_tiles = group_own_tiles(g->tiles_total, g->world, g->rank);
    hipSetDevice(c->cfg.device);
    int rc = ensure(c, g->send, (size_t)g->tiles_per_rank * 1024 * 4);
    if (!rc && g->rank == 0) rc = ensure(c, g->recv, (size_t)g->world * g->tiles_per_rank * 1024 * 4);
    // the padding tiles are never written by a render: zero them once, ordered before the first render on the context's own stream
    if (!rc) HIP_OK(c, hipMemsetAsync(g->send.p, 0, (size_t)g->tiles_per_rank * 1024 * 4, c->stream));
    return rc;
}

// The partition and the tile-major slot order as plain host functions (no GPU needed): what rank `rank` of `world` renders and where
// each of its pixels travels in the gather.  They call the very functions the kernels use (rtx_trace.h: group_*, tile_slot_pixel).
extern "C" int rtx_group_layout(int32_t width, int32_t height, int32_t world, int32_t rank, int32_t * tiles_total, int32_t * tiles_per_rank, int32_t * own_tiles) {
    if (width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world) return RTX_ERR_INVALID_ARG;
    const int tcx = (width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE, tcy = (height + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    if (tiles_total) *tiles_total = tcx * tcy;
    if (tiles_per_rank) *tiles_per_rank = group_tiles_per_rank(tcx * tcy, world);
    if (own_tiles) *own_tiles = group_own_tiles(tcx * tcy, world, rank);
    return RTX_OK;
}

// pixel_index[i] = raster index (y * width + x) of slot i of the rank's send buffer (tiles_per_rank * 1024 slots), or -1 for padding
// (clipped pixels of edge tiles, the unused tail when the tiles do not divide evenly).  Rank 0's receive buffer is the concatenation
// of every rank's send buffer in rank order, which is all k_unswizzle relies on.
extern "C" int rtx_group_slot_pixels(int32_t width, int32_t height, int32_t world, int32_t rank, int64_t * pixel_index, int64_t capacity) {
    int32_t total = 0, per_rank = 0, own = 0;
    const int rc = rtx_group_layout(width, height, world, rank, &total, &per_rank, &own);
    if (rc) return rc;
    if (!pixel_index || capacity < (int64_t)per_rank * 1024) return RTX_ERR_INVALID_ARG;
    const int tcx = (width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    for (int k = 0; k < per_rank; k++) {
        const int tile = group_tile_of(world, rank, k);
        for (int l = 0; l < 1024; l++) {
            int px, py; tile_slot_pixel(tile, l, tcx, px, py);
            pixel_index[(int64_t)k * 1024 + l] = (k < own && tile < total && px < width && py < height) ? (int64_t)py * width + px : -1;
        }
    }
    return RTX_OK;
}

extern "C" int rtx_group_unique_id(void * id128) {
    if (!id128) return RTX_ERR_INVALID_ARG;
    RcclApi * r = rccl();
    if (!r) return RTX_ERR_STATE;
    rtx_nccl_id id;
    if (r->GetUniqueId(&id) != 0) return RTX_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return RTX_OK;
}

extern "C" int rtx_group_create(rtx_ctx * c, int32_t rank, int32_t world, const void * id128, rtx_group ** out) {
    if (!c || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return RTX_ERR_INVALID_ARG;
    *out = nullptr;
    rtx_group * g = new rtx_group();
    g->ctx = c; g->rank = rank; g->world = world;
    RcclApi * r = rccl();
    if (!r) { c->err = "RCCL (librccl.so.1) could not be loaded"; delete g; return RTX_ERR_STATE; }
    hipSetDevice(c->cfg.device);
    rtx_nccl_id id; memset(&id, 0, sizeof(id));
    if (id128) memcpy(&id, id128, sizeof(id)); else if (r->GetUniqueId(&id) != 0) { delete g; return RTX_ERR_HIP; }
    g->hub = new GroupHub(); g->hub->device = c->cfg.device;
    const int e = r->CommInitRank(&g->hub->comm, world, id, rank);
    if (e != 0) { c->err = std::string("ncclCommInitRank: ") + r->GetErrorString(e); delete g->hub; delete g; return RTX_ERR_HIP; }
    const int rc = group_setup(g);
    if (rc) { r->CommDestroy(g->hub->comm); delete g->hub; delete g; return rc; }
    *out = g;
    return RTX_OK;
}

// A further context of the same process and GPU (another frame in flight) joins `base`'s communicator as the same rank.
extern "C" int rtx_group_attach(rtx_ctx * c, rtx_group * base, rtx_group ** out) {
    if (!c || !base || !base->hub || !out || c == base->ctx || c->cfg.device != base->hub->device ||
        c->cfg.width != base->ctx->cfg.width || c->cfg.height != base->ctx->cfg.height) return RTX_ERR_INVALID_ARG;
    *out = nullptr;
    hipSetDevice(c->cfg.device);
    GroupHub * h = base->hub;
    if (!h->xs) {      // the exchange stream gets the highest priority: RCCL's copy kernel needs a few workgroups' worth of room beside the frames' persistent grids, and should be first in line for it
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (knob_int("RTX_XS_PRIORITY", 1, 0, 1)) HIP_OK(c, hipStreamCreateWithPriority(&h->xs, hipStreamNonBlocking, hi));
        else HIP_OK(c, hipStreamCreateWithFlags(&h->xs, hipStreamNonBlocking));
    }
    rtx_group * g = new rtx_group();
    g->ctx = c; g->rank = base->rank; g->world = base->world; g->hub = h; h->refs++;
    const int rc = group_setup(g);
    if (rc) { h->refs--; delete g; return rc; }
    *out = g;
    return RTX_OK;
}

// One process driving n GPUs (the shape of the reference's own main loop): contexts[i] lives on its own device and becomes rank i.
extern "C" int rtx_group_create_local(rtx_ctx ** contexts, int32_t n, rtx_group ** out_groups) {
    if (!contexts || !out_groups || n < 1 || n > 64) return RTX_ERR_INVALID_ARG;
    RcclApi * r = rccl();
    if (!r) return RTX_ERR_STATE;
    std::vector<int> devs(n); std::vector<rtx_nccl_comm> comms(n, nullptr);
    for (int i = 0; i < n; i++) { if (!contexts[i]) return RTX_ERR_INVALID_ARG; devs[i] = contexts[i]->cfg.device; out_groups[i] = nullptr; }
    const int e = r->CommInitAll(comms.data(), n, devs.data());
    if (e != 0) { contexts[0]->err = std::string("ncclCommInitAll: ") + r->GetErrorString(e); return RTX_ERR_HIP; }
    int rc = RTX_OK;
    for (int i = 0; i < n; i++) {                                   // every communicator gets an owner first, so that one rollback path frees them all
        rtx_group * g = new rtx_group();
        g->ctx = contexts[i]; g->rank = i; g->world = n; g->hub = new GroupHub(); g->hub->comm = comms[i]; g->hub->device = devs[i];
        out_groups[i] = g;
    }
    for (int i = 0; i < n && !rc; i++) rc = group_setup(out_groups[i]);
    if (rc) { for (int i = 0; i < n; i++) { rtx_group_destroy(out_groups[i]); out_groups[i] = nullptr; } }
    return rc;
}

static int group_render_one(rtx_group * g, uint32_t flags, bool render, bool gather, bool finish) {
    rtx_ctx * c = g->ctx;
    RcclApi * r = rccl();
    hipSetDevice(c->cfg.device);
    if (render) {
        const int rc = render_tiles_impl(c, g->rank, g->world, g->own_tiles, flags | RTX_RENDER_SERIAL, (uint32_t *)g->send.p);
        if (rc) return rc;
    }
    if (gather) {
        // the one exchange of a frame: every rank's tiles_per_rank * 4 KiB of packed pixels to rank 0, on the stream the frame was rendered on
        hipStream_t xs = c->stream;
        if (g->hub->xs) {                                   // shared communicator: the exchange stream orders this rank's collectives
            if (!g->ev_rendered) { HIP_OK(c, hipEventCreateWithFlags(&g->ev_rendered, hipEventDisableTiming)); HIP_OK(c, hipEventCreateWithFlags(&g->ev_gathered, hipEventDisableTiming)); }
            xs = g->hub->xs;
            HIP_OK(c, hipEventRecord(g->ev_rendered, c->stream)); HIP_OK(c, hipStreamWaitEvent(xs, g->ev_rendered, 0));
        }
        const int e = r->Gather(g->send.p, g->recv.p, (size_t)g->tiles_per_rank * 1024, /*ncclUint32*/ 3, 0, g->hub->comm, xs);
        if (e != 0) { c->err = std::string("ncclGather: ") + r->GetErrorString(e); return RTX_ERR_HIP; }
        if (xs != c->stream) { HIP_OK(c, hipEventRecord(g->ev_gathered, xs)); HIP_OK(c, hipStreamWaitEvent(c->stream, g->ev_gathered, 0)); }
    }
    if (finish && g->rank == 0) {
        const int tcx = (c->cfg.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
        hipLaunchKernelGGL(k_unswizzle, dim3(c->n_cu * 4), dim3(256), 0, c->stream, (const uint32_t *)g->recv.p,
                           (uint32_t *)(c->ext_packed ? c->ext_packed : c->d_fb_packed.p), g->world, g->tiles_per_rank, g->tiles_total, tcx, c->cfg.width, c->cfg.height);
        HIP_OK(c, hipGetLastError());
    }
    return RTX_OK;
}

// Renders this rank's tile shard (tile t belongs to rank t mod world, WorkerThread.cpp:53-65 without the atomic counter), gathers the
// packed tiles to rank 0 and, there, writes the frame into the context's packed framebuffer.  Everything is queued on the context's
// stream; nothing synchronises with the host.
extern "C" int rtx_group_render(rtx_group * g, uint32_t flags) {
    if (!g || !g->ctx) return RTX_ERR_INVALID_ARG;
    return group_render_one(g, flags, true, true, true);
}

extern "C" int rtx_group_render_local(rtx_group ** groups, int32_t n, uint32_t flags) {
    if (!groups || n < 1) return RTX_ERR_INVALID_ARG;
    RcclApi * r = rccl();
    if (!r) return RTX_ERR_STATE;
    for (int i = 0; i < n; i++) { if (!groups[i]) return RTX_ERR_INVALID_ARG; const int rc = group_render_one(groups[i], flags, true, false, false); if (rc) return rc; }
    r->GroupStart();                                               // one thread issues every rank's call of the collective
    int rc = RTX_OK;
    for (int i = 0; i < n && !rc; i++) rc = group_render_one(groups[i], flags, false, true, false);
    r->GroupEnd();
    for (int i = 0; i < n && !rc; i++) rc = group_render_one(groups[i], flags, false, false, true);
    return rc;
}

extern "C" int rtx_group_destroy(rtx_group * g) {
    if (!g) return RTX_ERR_INVALID_ARG;
    if (g->ctx) { hipSetDevice(g->ctx->cfg.device); if (g->ctx->stream) hipStreamSynchronize(g->ctx->stream); }
    if (g->hub) {
        if (g->hub->xs) hipStreamSynchronize(g->hub->xs);
        if (--g->hub->refs == 0) { RcclApi * r = rccl(); if (r && g->hub->comm) r->CommDestroy(g->hub->comm); if (g->hub->xs) hipStreamDestroy(g->hub->xs); delete g->hub; }
    }
    if (g->ev_rendered) hipEventDestroy(g->ev_rendered);
    if (g->ev_gathered) hipEventDestroy(g->ev_gathered);
    if (g->send.p) hipFree(g->send.p);
    if (g->recv.p) hipFree(g->recv.p);
    delete g;
    return RTX_OK;
}

// Test hook: the whole group path of `world` ranks on ONE GPU without RCCL — every rank's tile shard rendered tile-major straight into
// its slice of rank 0's receive buffer, then the same k_unswizzle.  Covers the partition, the tile-major writes and the frame assembly
// for any world size; the ncclGather itself is exercised by rtx_group_render.
extern "C" int rtx_debug_group_loopback(rtx_ctx * c, int32_t world, uint32_t flags) {
    if (!c || world < 1) return RTX_ERR_INVALID_ARG;
    rtx_group g; g.ctx = c; g.rank = 0; g.world = world;
    int rc = group_setup(&g);
    for (int r = 0; r < world && !rc; r++) {
        const int own = group_own_tiles(g.tiles_total, world, r);
        rc = render_tiles_impl(c, r, world, own, flags | RTX_RENDER_SERIAL, (uint32_t *)g.recv.p + (size_t)r * g.tiles_per_rank * 1024);
    }
    if (!rc) {
        const int tcx = (c->cfg.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
        hipLaunchKernelGGL(k_unswizzle, dim3(c->n_cu * 4), dim3(256), 0, c->stream, (const uint32_t *)g.recv.p,
                           (uint32_t *)(c->ext_packed ? c->ext_packed : c->d_fb_packed.p), world, g.tiles_per_rank, g.tiles_total, tcx, c->cfg.width, c->cfg.height);
        hipStreamSynchronize(c->stream);
    }
    if (g.send.p) hipFree(g.send.p);
    if (g.recv.p) hipFree(g.recv.p);
    return rc;
}

// ---- unit-level entry points, second set: the traversal / hit / light / plot functions at caller-supplied inputs -------------------
// rtx_debug_trace_rays: n rays (18 floats: origin, direction, dO_dx, dO_dy, dD_dx, dD_dy) through Scene::trace_primitives as the frame
// kernels do it (the level-1 ray queue, the production closest-hit kernel, the accept-branch rebuild of k_shade), all RayHit fields out
// (27 floats, layout of oracle orc_trace_closest); rtx_debug_occluded: Scene::intersect_primitives through the production shadow-ray
// kernel (the level-0 shadow queue of light 0).
__global__ void k_debug_fill_rays(DevQueues q, const float * rays, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { q.counters->ray_count[1] = (uint32_t)n; }
    if (i >= n) return;
    const float * r = rays + 18 * (size_t)i;
    const uint32_t s = (uint32_t)q.level_base[1] + (uint32_t)i;
    q.r0[s] = make_float4(r[0], r[1], r[2], r[3]); q.r1[s] = make_float4(r[4], r[5], as_f(0), as_f(-1));      // (origin, d.x) (d.y, d.z, pixel, parent)
    q.r2[s] = make_float4(r[8], r[9], r[10], r[11]); q.r3[s] = make_float4(r[12], r[13], r[14], r[15]);
    q.r4[s] = make_float4(r[16], r[17], r[6], r[7]);
}
__global__ void k_debug_rebuild(DevScene sc, DevQueues q, const float * rays, float * out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float * r = rays + 18 * (size_t)i;
    RayFull ray;
    ray.o = V3(r[0], r[1], r[2]); ray.d = V3(r[3], r[4], r[5]); ray.dO_dx = V3(r[6], r[7], r[8]); ray.dO_dy = V3(r[9], r[10], r[11]);
    ray.dD_dx = V3(r[12], r[13], r[14]); ray.dD_dy = V3(r[15], r[16], r[17]);
    const uint32_t s = (uint32_t)q.level_base[1] + (uint32_t)i;
    const float4 h0 = q.h0[s]; const int tri = q.h1[s], prim = as_i(h0.w);
    float * o = out + 27 * (size_t)i;
    for (int k = 0; k < 27; k++) o[k] = 0.0f;
    o[1] = h0.x;
    if (PRIM_KIND(prim) == PRIM_NONE) return;
    HitFull h; h.material_id = 0;
    h.point = h.normal = h.dO_dx = h.dO_dy = h.dN_dx = h.dN_dy = V3(0, 0, 0); h.u = h.v = h.ds_dx = h.ds_dy = h.dt_dx = h.dt_dy = 0.0f;
    if (PRIM_KIND(prim) == PRIM_TRI) rebuild_triangle_hit(sc, PRIM_INDEX(prim), tri, h0.x, h0.y, h0.z, ray, h);
    else if (PRIM_KIND(prim) == PRIM_SPHERE) rebuild_sphere_hit(sc, sc.spheres[PRIM_INDEX(prim)], h0.x, ray, h);
    else rebuild_plane_hit(sc, sc.planes[PRIM_INDEX(prim)], h0.x, ray, h);
    o[0] = 1.0f; o[2] = h.point.x; o[3] = h.point.y; o[4] = h.point.z; o[5] = h.normal.x; o[6] = h.normal.y; o[7] = h.normal.z;
    o[8] = (float)h.material_id; o[9] = h.u; o[10] = h.v; o[11] = h.ds_dx; o[12] = h.ds_dy; o[13] = h.dt_dx; o[14] = h.dt_dy;
    o[15] = h.dO_dx.x; o[16] = h.dO_dx.y; o[17] = h.dO_dx.z; o[18] = h.dO_dy.x; o[19] = h.dO_dy.y; o[20] = h.dO_dy.z;
    o[21] = h.dN_dx.x; o[22] = h.dN_dx.y; o[23] = h.dN_dx.z; o[24] = h.dN_dy.x; o[25] = h.dN_dy.y; o[26] = h.dN_dy.z;
}
__global__ void k_debug_fill_shadow(DevQueues q, int nL, const float * od7, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) q.counters->ray_count[0] = (uint32_t)n;
    if (i >= n) return;
    const float * r = od7 + 7 * (size_t)i;
    q.n0[(uint32_t)q.level_base[0] + (uint32_t)i] = make_float4(0.0f, 0.0f, 0.0f, as_f(NF_HIT | NF_LIT));
    for (int L = 0; L < nL; L++) {
        const uint32_t s = (uint32_t)q.shadow_base[0] + (uint32_t)L * (uint32_t)q.level_cap[0] + (uint32_t)i;
        q.s0[s] = make_float4(r[0], r[1], r[2], L == 0 ? r[6] : -1.0f);      // only light 0's segment carries rays; the others are marked culled
        q.s1[s] = make_float4(r[3], r[4], r[5], 0.0f);
        q.socc[s] = 0u;
    }
}

static int debug_queues(rtx_ctx * c, int n) {
    if (!c->frame_set) { c->err = "debug trace before rtx_set_frame"; return RTX_ERR_STATE; }
    if (int bad = validate_references(c)) return bad;
    const int tiles = (n + 1023) / 1024;
    if (tiles > 4096) return RTX_ERR_LIMIT;
    int rc = alloc_queues(c, tiles < 1 ? 1 : tiles);
    if (rc) return rc;
    hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, c->stream, (DevCounters *)c->d_counters.p, (uint32_t *)c->d_pk_heads.p, 0u,
                       (uint32_t)(c->trace_blocks_closest * RTX_TRACE_BLOCK), (uint32_t)(c->trace_blocks_any * RTX_TRACE_BLOCK),
                       (uint32_t)(c->pk_blocks_closest * (RTX_PK_BLOCK / RTX_WAVE)), (uint32_t)(c->pk_blocks_any * (RTX_PK_BLOCK / RTX_WAVE)), 1);
    return RTX_OK;
}

extern "C" int rtx_debug_trace_rays(rtx_ctx * c, const float * rays18, int32_t n, float * hits27, uint32_t flags) {
    if (!c || !rays18 || !hits27 || n <= 0 || c->cfg.bounces < 1) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    int rc = debug_queues(c, n);
    if (rc) return rc;
    float * din = nullptr, * dout = nullptr;
    HIP_OK(c, hipMalloc(&din, (size_t)n * 72)); HIP_OK(c, hipMalloc(&dout, (size_t)n * 108));
    HIP_OK(c, hipMemcpy(din, rays18, (size_t)n * 72, hipMemcpyHostToDevice));
    DevQueues q = c->q; q.first_tile = 0; q.tile_stride = 1; q.tile_count = (n + 1023) / 1024; q.primary_slots = 0; q.tm_packed = nullptr;
    const DevScene sc = c->scene;
    hipLaunchKernelGGL(k_debug_fill_rays, dim3((n + 255) / 256), dim3(256), 0, c->stream, q, din, n);
    if (flags & RTX_RENDER_LANE_TRACE) hipLaunchKernelGGL((k_trace_fast<false>), dim3(c->trace_blocks_closest), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, 1, 1);
    else {
        if (flags & RTX_RENDER_PACKET_CLOSEST) q.pk_defer_t0_closest = q.pk_defer_t0_primary = q.pk_defer_t0;
        hipLaunchKernelGGL((k_packet<false, false>), dim3(c->pk_blocks_closest), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, 1, 1);
    }
    hipLaunchKernelGGL(k_debug_rebuild, dim3((n + 255) / 256), dim3(256), 0, c->stream, sc, q, din, dout, n);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(hits27, dout, (size_t)n * 108, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout);
    return RTX_OK;
}

extern "C" int rtx_debug_occluded(rtx_ctx * c, const float * origin_direction_maxdist7, int32_t n, uint32_t * occluded, uint32_t flags) {
    if (!c || !origin_direction_maxdist7 || !occluded || n <= 0 || c->scene.light_count < 1) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    int rc = debug_queues(c, n);
    if (rc) return rc;
    float * din = nullptr;
    HIP_OK(c, hipMalloc(&din, (size_t)n * 28));
    HIP_OK(c, hipMemcpy(din, origin_direction_maxdist7, (size_t)n * 28, hipMemcpyHostToDevice));
    DevQueues q = c->q; q.first_tile = 0; q.tile_stride = 1; q.tile_count = (n + 1023) / 1024; q.primary_slots = n; q.tm_packed = nullptr;
    {   // explicit shadow rays (the frame kernels derive theirs from the hit's point and the light): origin + distance, direction per shadow slot of level 0
        const size_t sslots = (size_t)q.level_cap[0] * (size_t)(c->scene.light_count > 0 ? c->scene.light_count : 1);
        rc = ensure(c, c->d_dbg_s0, sslots * 16); if (!rc) rc = ensure(c, c->d_dbg_s1, sslots * 16);
        if (rc) { hipFree(din); return rc; }
        q.s0 = (float4 *)c->d_dbg_s0.p; q.s1 = (float4 *)c->d_dbg_s1.p; q.shadow_explicit = 1;
    }
    const DevScene sc = c->scene;
    hipLaunchKernelGGL(k_debug_fill_shadow, dim3((n + 255) / 256), dim3(256), 0, c->stream, q, sc.light_count, din, n);
    if (flags & RTX_RENDER_LANE_TRACE) hipLaunchKernelGGL((k_trace_fast<true>), dim3(c->trace_blocks_any), dim3(RTX_TRACE_BLOCK), 0, c->stream, sc, q, 0, 0);
    else hipLaunchKernelGGL((k_packet<true, false>), dim3(c->pk_blocks_any), dim3(RTX_PK_BLOCK), 0, c->stream, sc, q, 0, 0);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(occluded, q.socc + q.shadow_base[0], (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(din);
    return RTX_OK;
}

__global__ void k_debug_light_plot(rtx_point_light pl, rtx_spot_light sl, rtx_directional_light dl, const float * in10, float * out9, const float * rgb, uint32_t * packed, int n_light, int n_plot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_light) {
        const float * r = in10 + 10 * (size_t)i;
        const v3 nrm = V3(r[0], r[1], r[2]), tl = V3(r[3], r[4], r[5]), tc = V3(r[6], r[7], r[8]); const float d2 = r[9];
        const v3 a = vdivs(light_calc(v3p(pl.colour), nrm, tl, tc), d2);                      // PointLight.h:9-11
        v3 b = V3(0.0f, 0.0f, 0.0f);                                                           // SpotLight.h:17-33, as k_shade evaluates it
        const float dt = vdot(tl, v3p(sl.negative_direction));
        if (dt > sl.outer_cutoff) { float f = (dt - sl.outer_cutoff) / (sl.inner_cutoff - sl.outer_cutoff); f = (f > 1.0f) ? 1.0f : f; b = vmuls(vdivs(light_calc(v3p(sl.colour), nrm, tl, tc), d2), f); }
        const v3 cc = light_calc(v3p(dl.colour), nrm, v3p(dl.negative_direction), tc);         // DirectionalLight.h:9-11
        float * o = out9 + 9 * (size_t)i;
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = b.x; o[4] = b.y; o[5] = b.z; o[6] = cc.x; o[7] = cc.y; o[8] = cc.z;
    }
    if (i < n_plot) packed[i] = plot_pack(V3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));     // Window::plot, Window.h:56-65
}

// Light::calc_lighting of the three light kinds (k_shade's expressions) and Window::plot at caller-supplied inputs
extern "C" int rtx_debug_light_plot(rtx_ctx * c, const rtx_point_light * pl, const rtx_spot_light * sl, const rtx_directional_light * dl,
                                    const float * in10, float * out9, int32_t n_light, const float * rgb, uint32_t * packed, int32_t n_plot) {
    if (!c || !pl || !sl || !dl || n_light < 0 || n_plot < 0 || (n_light && (!in10 || !out9)) || (n_plot && (!rgb || !packed))) return RTX_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    float * din = nullptr, * dout = nullptr, * drgb = nullptr; uint32_t * dpk = nullptr;
    HIP_OK(c, hipMalloc(&din, (size_t)(n_light + 1) * 40)); HIP_OK(c, hipMalloc(&dout, (size_t)(n_light + 1) * 36));
    HIP_OK(c, hipMalloc(&drgb, (size_t)(n_plot + 1) * 12)); HIP_OK(c, hipMalloc(&dpk, (size_t)(n_plot + 1) * 4));
    if (n_light) HIP_OK(c, hipMemcpy(din, in10, (size_t)n_light * 40, hipMemcpyHostToDevice));
    if (n_plot) HIP_OK(c, hipMemcpy(drgb, rgb, (size_t)n_plot * 12, hipMemcpyHostToDevice));
    const int n = n_light > n_plot ? n_light : n_plot;
    hipLaunchKernelGGL(k_debug_light_plot, dim3((n + 255) / 256), dim3(256), 0, c->stream, *pl, *sl, *dl, din, dout, drgb, dpk, n_light, n_plot);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (n_light) HIP_OK(c, hipMemcpy(out9, dout, (size_t)n_light * 36, hipMemcpyDeviceToHost));
    if (n_plot) HIP_OK(c, hipMemcpy(packed, dpk, (size_t)n_plot * 4, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout); hipFree(drgb); hipFree(dpk);
    return RTX_OK;
}
