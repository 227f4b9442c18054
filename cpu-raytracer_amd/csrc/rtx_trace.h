// rtx_trace.h — the traversal kernels: closest hit (Scene::trace_primitives) and any hit
// (Scene::intersect_primitives) as ONE persistent-threads kernel template.
//
// Structure (MI355X): a fixed grid of 256-thread workgroups stays resident; every lane owns one
// ray and a private traversal stack striped through LDS (entry e of lane l at [e][l]: conflict
// free).  When at least RTX_REFILL_MIN lanes of a wave have retired their ray the wave refills
// them from the HBM ray queue: a 64-bit ballot of idle lanes, one atomicAdd on the queue head by
// the leader, and a popcount-prefix rank per lane (dynamic fetch).  The two-level walk
// (TopLevelBVH -> Mesh -> BottomLevelBVH) runs on one stack: TLAS leaves push instance entries
// (negative), an instance entry switches the lane to model space until the stack drops back to the
// level it had on entry.
//
// Per-lane traversal ORDER is the reference's (ordered by split axis vs ray direction sign,
// BVHNode.h:30-40; far child pushed first, TopLevelBVH.cpp:76-82, BottomLevelBVH.cpp:381-387) so
// that equal-t ties and ulp-level culling resolve identically.
#pragma once
#include "rtx_math.h"
#include "rtx_device.h"

#ifndef RTX_REFILL_MIN
#define RTX_REFILL_MIN   48     // refill a wave when this many lanes are idle (A/B on MI355X: 16 -> 5.57 ms, 48 -> 4.39 ms per cfg3 frame)
#endif
#ifndef RTX_TRACE_STEPS
#define RTX_TRACE_STEPS  32     // traversal steps between refill checks
#endif
#ifndef RTX_INNER_REPEAT
#define RTX_INNER_REPEAT 1      // inner-node steps per traversal iteration (while-while style convergence)
#endif
#ifndef RTX_INNER_MIN_LANES
#define RTX_INNER_MIN_LANES 32  // keep repeating the inner step while at least this many lanes want it
#endif
#ifndef RTX_WAVES_ANY
#define RTX_WAVES_ANY     1     // __launch_bounds__ min waves per SIMD for the shadow-ray kernel
#endif
#ifndef RTX_WAVES_CLOSEST
#define RTX_WAVES_CLOSEST 1
#endif
#define RAY_EPSILON      0.005f // Ray.h:5

// Per-lane pointers to BVH nodes / triangles come out of device tables (DevBlas), so the compiler cannot prove their address
// space and would emit FLAT loads (LDS-aperture check, lgkmcnt + vmcnt).  They always point into HBM: say so.
#ifndef RTX_NANFREE_ANY
#define RTX_NANFREE_ANY 1        // use the NaN-free slab test in waves without +-0 direction components (shadow-ray kernel)
#endif
#ifndef RTX_NANFREE_CLOSEST
#define RTX_NANFREE_CLOSEST 0    // same for the closest-hit kernel (costs it registers: 119 -> 139 VGPRs, one wave per SIMD less)
#endif

// Tile-major slot order, shared by the primary-ray set-up, k_resolve's tile-major writes, k_unswizzle and the host-side
// rtx_group_slot_pixels (one definition: host and device cannot disagree).  1024 slots per tile, slot l = 8x8 pixel block l >> 6
// (row-major blocks of 4 x 4 per tile), pixel l & 63 inside it (row-major): a wave covers one block.
__host__ __device__ __forceinline__ void tile_slot_pixel(int tile, int l, int tile_count_x, int & px, int & py) {
    const int tx = (tile % tile_count_x) * RTX_TILE_SIZE;    // WorkerThread.cpp:57-58
    const int ty = (tile / tile_count_x) * RTX_TILE_SIZE;
    const int blk = l >> 6, ln = l & 63;
    px = tx + (blk & 3) * 8 + (ln & 7);
    py = ty + (blk >> 2) * 8 + (ln >> 3);
}
// GPU group partition: rank r of `world` owns tiles r, r + world, ...; every rank sends tiles_per_rank tiles (padded)
__host__ __device__ __forceinline__ int group_tiles_per_rank(int tiles_total, int world) { return (tiles_total + world - 1) / world; }
__host__ __device__ __forceinline__ int group_own_tiles(int tiles_total, int world, int rank) { return rank < tiles_total ? (tiles_total - rank + world - 1) / world : 0; }
__host__ __device__ __forceinline__ int group_tile_of(int world, int rank, int k) { return rank + k * world; }      // the rank's k-th tile

// Slot -> pixel of a primary ray.
RTX_D bool primary_pixel(const DevScene & sc, const DevQueues & q, uint32_t slot, int & px, int & py) {
    const int tile_k = slot >> 10, l = slot & 1023;
    tile_slot_pixel(q.first_tile + tile_k * q.tile_stride, l, sc.tile_count_x, px, py);
    return px < sc.width && py < sc.height;                  // WorkerThread.cpp:60-61 (clipped tiles)
}

struct RayFull {
    v3 o, d, dO_dx, dO_dy, dD_dx, dD_dy;
};

// Raytracer::render_tile, Raytracer.cpp:30-59 (lane 1)
RTX_D void primary_ray(const DevScene & sc, int px, int py, RayFull & r, bool with_differentials) {
    v3 ax = v3p(sc.cam_x), ay = v3p(sc.cam_y), tl = v3p(sc.cam_tl);
    float is = (float)px, js = (float)py;
    v3 d = vmadd_s(ax, is, vmadd_s(ay, js, tl));
    float dd  = vdot(d, d);
    float inv = 1.0f / sqrtf(dd);
    r.o = v3p(sc.cam_pos);
    r.d = vmuls(d, inv);
    if (with_differentials) {
        float denom = inv / dd;
        r.dO_dx = V3(0.0f, 0.0f, 0.0f);
        r.dO_dy = V3(0.0f, 0.0f, 0.0f);
        r.dD_dx = vmuls(vsub(vmuls(ax, dd), vmuls(d, vdot(d, ax))), denom);
        r.dD_dy = vmuls(vsub(vmuls(ay, dd), vmuls(d, vdot(d, ay))), denom);
    }
}

// AABB::intersect, AABB.cpp:38-52
RTX_D bool slab_test(float4 a, float4 b, v3 o, v3 inv, float max_distance) {
    v3 t0 = vmul(vsub(V3(a.x, a.y, a.z), o), inv);
    v3 t1 = vmul(vsub(V3(b.x, b.y, b.z), o), inv);
    v3 tmin = V3(fmin_ref(t0.x, t1.x), fmin_ref(t0.y, t1.y), fmin_ref(t0.z, t1.z));
    v3 tmax = V3(fmax_ref(t0.x, t1.x), fmax_ref(t0.y, t1.y), fmax_ref(t0.z, t1.z));
    float t_near = fmax_ref(fmax_ref(RAY_EPSILON, tmin.x), fmax_ref(tmin.y, tmin.z));
    float t_far  = fmin_ref(fmin_ref(max_distance, tmax.x), fmin_ref(tmax.y, tmax.z));
    return t_near < t_far;
}

// The same test, also returning the key that decides the reference's POP-time re-test of this node.  With min(a,b) = a<b?a:b,
// t_far(D) = min(min(D, X), M), X = tmax.x, M = min(tmax.y, tmax.z).  If the node passes for some D, then for any later D' <= D:
//   X is NaN (0 * inf on the x slab): min(D, NaN) = NaN and min(NaN, M) = M, so t_far does not depend on D -> it passes again;
//   otherwise t_near < X and t_near < M are already known, so pass(D') == (t_near < D').
// key = -inf in the first case, t_near in the second: the pop-time test is exactly `key < D'`, with no memory access.
RTX_D bool slab_test_key(float4 a, float4 b, v3 o, v3 inv, float max_distance, float & key) {
    v3 t0 = vmul(vsub(V3(a.x, a.y, a.z), o), inv);
    v3 t1 = vmul(vsub(V3(b.x, b.y, b.z), o), inv);
    v3 tmin = V3(fmin_ref(t0.x, t1.x), fmin_ref(t0.y, t1.y), fmin_ref(t0.z, t1.z));
    v3 tmax = V3(fmax_ref(t0.x, t1.x), fmax_ref(t0.y, t1.y), fmax_ref(t0.z, t1.z));
    float t_near = fmax_ref(fmax_ref(RAY_EPSILON, tmin.x), fmax_ref(tmin.y, tmin.z));
    float t_far  = fmin_ref(fmin_ref(max_distance, tmax.x), fmin_ref(tmax.y, tmax.z));
    key = (tmax.x != tmax.x) ? -INFINITY : t_near;
    return t_near < t_far;
}

// NaN-free variant.  The only NaN a slab test can produce is 0 * inf: a box plane through the ray origin on an axis whose
// direction component is +-0 (inverse +-inf).  When all three inverse components are FINITE every t0 / t1 is finite or +-inf,
// and on non-NaN operands `a < b ? a : b` and the hardware v_min_f32 / v_max_f32 / v_min3 / v_max3 agree (up to the sign of a
// zero, which no comparison below can see: t_near >= EPSILON > 0).  Half the VALU instructions of the reference-form test.
// FAST must only be true when inv.x, inv.y, inv.z are all finite (the kernels check per wave).
template <bool FAST>
RTX_D bool slab_test_sel(float4 a, float4 b, v3 o, v3 inv, float max_distance, float & key) {
    if (!FAST) return slab_test_key(a, b, o, inv, max_distance, key);
    v3 t0 = vmul(vsub(V3(a.x, a.y, a.z), o), inv);
    v3 t1 = vmul(vsub(V3(b.x, b.y, b.z), o), inv);
    const float t_near = __builtin_fmaxf(__builtin_fmaxf(RAY_EPSILON, __builtin_fminf(t0.x, t1.x)), __builtin_fmaxf(__builtin_fminf(t0.y, t1.y), __builtin_fminf(t0.z, t1.z)));
    const float t_far  = __builtin_fminf(__builtin_fminf(max_distance, __builtin_fmaxf(t0.x, t1.x)), __builtin_fminf(__builtin_fmaxf(t0.y, t1.y), __builtin_fmaxf(t0.z, t1.z)));
    key = t_near;
    return t_near < t_far;
}
RTX_D bool inv_is_finite(v3 inv) {
    return (__builtin_fabsf(inv.x) < INFINITY) & (__builtin_fabsf(inv.y) < INFINITY) & (__builtin_fabsf(inv.z) < INFINITY);   // false for inf and NaN
}

// ---- which rays need the reference-form slab test ---------------------------------------------------------------------------------------
// The fast forms (hardware min / max, the sign-coherent test, the nesting arguments of the 4-wide records) equal the reference's `a < b ? a : b`
// forms as long as no slab test produces a NaN, and the only NaN a slab test can produce is 0 * inf: an infinite inverse-direction component
// (a direction component of +-0, or so small that its reciprocal overflows) times a box plane that passes exactly through the ray's origin on that
// axis.  A zero component alone is harmless — (b - o) * inf is +-inf, ordered like any other value, and AABB::intersect's comparisons see no
// difference — and common: the middle row of pixels of any camera that is only rotated about y has d.y == 0.  So a ray is sent to the
// reference-form walker only if, on an axis with a non-finite inverse, its origin equals one of the tree's box-plane coordinates (the distinct
// values per axis, sorted at upload: a binary search by the few lanes concerned).
RTX_D bool plane_member(const float * __restrict__ planes, const int n, const float x) {
    if (x != x) return true;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (planes[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && planes[lo] == x;
}
RTX_D bool pk_nan_possible(const DevBlas & B, const v3 o, const v3 inv) {
    bool r = false;
    if (!(__builtin_fabsf(inv.x) < INFINITY)) r |= (inv.x != inv.x) || plane_member(B.planes[0], B.plane_count[0], o.x);
    if (!(__builtin_fabsf(inv.y) < INFINITY)) r |= (inv.y != inv.y) || plane_member(B.planes[1], B.plane_count[1], o.y);
    if (!(__builtin_fabsf(inv.z) < INFINITY)) r |= (inv.z != inv.z) || plane_member(B.planes[2], B.plane_count[2], o.z);
    return r;
}
// the same question for the TLAS (a handful of nodes, rebuilt every frame: its planes are compared one by one; a large TLAS is not searched)
RTX_D bool pk_nan_possible_tlas(const DevScene & sc, const v3 o, const v3 inv) {
    const bool ix = !(__builtin_fabsf(inv.x) < INFINITY), iy = !(__builtin_fabsf(inv.y) < INFINITY), iz = !(__builtin_fabsf(inv.z) < INFINITY);
    if (!(ix | iy | iz)) return false;
    if (sc.tlas_node_count > 128 || inv.x != inv.x || inv.y != inv.y || inv.z != inv.z || o.x != o.x || o.y != o.y || o.z != o.z) return true;
    bool r = false;
    for (int i = 0; i < sc.tlas_node_count; i++) {
        const float4 a = sc.tlas_nodes[2 * i], b = sc.tlas_nodes[2 * i + 1];          // (min.xyz, -) (max.xyz, -)
        r |= (ix & ((a.x == o.x) | (b.x == o.x))) | (iy & ((a.y == o.y) | (b.y == o.y))) | (iz & ((a.z == o.z) | (b.z == o.z)));
    }
    return r;
}


// Sphere::trace hit test only (Sphere.cpp:9-42); attributes are rebuilt in the shade kernel
RTX_D bool sphere_closest(const rtx_sphere & sp, v3 o, v3 d, float tbest, float & t_out) {
    v3 oc = vsub(o, v3p(sp.center));
    float a = vdot(d, d);
    float b = 2.0f * vdot(oc, d);
    float c = vdot(oc, oc) - sp.radius_squared;
    float disc = b * b - 4.0f * a * c;
    if (!(disc >= 0.0f)) return false;
    float sqrt_d    = sqrtf(disc);
    float inv_denom = -(1.0f / (2.0f * a));
    float t0 = (b + sqrt_d) * inv_denom;
    float t1 = (b - sqrt_d) * inv_denom;
    float t = (t0 > RAY_EPSILON) ? t0 : t1;
    if (!((t > RAY_EPSILON) & (t < tbest))) return false;
    t_out = t;
    return true;
}
// Sphere::intersect, Sphere.cpp:92-112
RTX_D bool sphere_any(const rtx_sphere & sp, v3 o, v3 d, float max_distance) {
    v3 c = vsub(v3p(sp.center), o);
    float t = vdot(c, d);
    v3 Q = vsub(c, vmuls(d, t));
    float p2 = vdot(Q, Q);
    float rs = sp.radius_squared;
    if (!(p2 < rs)) return false;
    t = t - sqrtf(rs - p2);
    return (t > RAY_EPSILON) & (t < max_distance);
}
// Plane::trace / Plane::intersect distance, Plane.cpp:18,75
RTX_D float plane_t(const rtx_plane & pl, v3 o, v3 d) {
    v3 n = v3p(pl.normal);
    return -(vdot(n, o) + pl.distance) / vdot(n, d);
}

// Shadow queue of a level: one slot per ray of the level and light (slot = the ray's own index: k_shade does not compact), of which
// only the rays with a lit hit carry a shadow ray — NF_LIT in the node record k_shade wrote.
// The shadow ray of light L (point lights, then spot lights, then directional ones: the order of Raytracer.cpp:155-198) for a hit at `point`:
// direction to the light and the distance the ray may travel.  ONE definition for every consumer: the bits of a derived ray do not depend on who derives it.
RTX_D void shadow_ray_of_light(const DevScene & sc, const int L, const v3 point, v3 & dir, float & dist) {
    if (L < sc.point_light_count) {
        v3 to_light = vsub(v3p(sc.point_lights[L].position), point);
        const float d2 = vdot(to_light, to_light);
        const float d  = sqrtf(d2);
        dir = vdivs(to_light, d); dist = d;
    } else if (L < sc.point_light_count + sc.spot_light_count) {
        v3 to_light = vsub(v3p(sc.spot_lights[L - sc.point_light_count].position), point);
        const float d2 = vdot(to_light, to_light);
        const float d  = sqrtf(d2);
        dir = vdivs(to_light, d); dist = d;
    } else {
        dir = v3p(sc.dir_lights[L - sc.point_light_count - sc.spot_light_count].negative_direction); dist = INFINITY;
    }
}
// origin, direction and maximum distance of the shadow ray in shadow slot `out_slot` = (level, light, rank); a negative distance = not to be traced
RTX_D void load_shadow_ray(const DevScene & sc, const DevQueues & q, const int level, const int light, const uint32_t rank, const uint32_t out_slot, v3 & o, v3 & d, float & tmax) {
    if (q.shadow_explicit) { const float4 s0 = q.s0[out_slot], s1 = q.s1[out_slot]; o = V3(s0.x, s0.y, s0.z); tmax = s0.w; d = V3(s1.x, s1.y, s1.z); return; }
    const float4 p = q.sp[(uint32_t)q.level_base[level] + rank];
    o = V3(p.x, p.y, p.z);
    shadow_ray_of_light(sc, light, o, d, tmax);
    if (q.cull && q.socc[out_slot] != 0u) tmax = -1.0f;          // k_shade<.., CULL>: this light contributes exactly nothing to this hit
}

RTX_D uint32_t shadow_slots(const DevCounters * ctr, const DevQueues & q, int level) {
    const uint32_t n = ctr->ray_count[level], cap = (uint32_t)q.level_cap[level];
    return n < cap ? n : cap;
}
RTX_D bool shadow_slot_is_lit(const DevQueues & q, int level, uint32_t rank) {
    return (as_i(q.n0[(uint32_t)q.level_base[level] + rank].w) & NF_LIT) != 0;
}

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(RTX_TRACE_BLOCK)
void k_trace(const DevScene sc, const DevQueues q, const int level) {
    __shared__ int lds_stack[RTX_TRACE_BLOCK / RTX_WAVE][RTX_LDS_STACK][RTX_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int * const stk = &lds_stack[wave][0][lane];
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    int * const spill = q.spill + gtid;
    const int spill_stride = q.spill_threads;

    DevCounters * const ctr = q.counters;
    const uint32_t total = ANY ? shadow_slots(ctr, q, level) * (uint32_t)sc.light_count : ctr->ray_count[level];
    uint32_t * const head = ANY ? &ctr->fetch_any[level] : &ctr->fetch_closest[level];
    const int lit_cap = q.level_cap[level];
    const uint32_t lit_n = shadow_slots(ctr, q, level);

    // per-lane ray state
    bool has_ray = false;
    uint32_t out_slot = 0;
    v3 wo = V3(0, 0, 0), wd = wo, winv = wo;       // world-space ray
    v3 co = wo, cd = wo, cinv = wo;                // ray in the space being traversed
    float tcur = 0.0f;                             // closest t so far (ANY: max distance)
    int best_prim = PRIM_NONE, best_tri = 0; float best_u = 0.0f, best_v = 0.0f;
    int sp = 0, floor_sp = -1, cur_inst = 0;
    rtx_gptr cur_nodes = RTX_GPTR(sc.tlas_nodes);
    rtx_gptr cur_tris  = nullptr;
    uint32_t n_tlas = 0, n_blas = 0, n_inst = 0, n_tri = 0, n_rays = 0, n_stackerr = 0;
    int bvh_steps = 0;                             // RayHit::bvh_steps (RayHit.h:24-26), kept only for sc.heatmap

    auto push = [&](int e) {
        if (sp < RTX_LDS_STACK) stk[sp * RTX_WAVE] = e;
        else if (sp < RTX_MAX_STACK) spill[(sp - RTX_LDS_STACK) * spill_stride] = e;
        else { n_stackerr = 1; return; }
        sp++;
    };
    auto pop = [&]() -> int {
        sp--;
        return (sp < RTX_LDS_STACK) ? stk[sp * RTX_WAVE] : spill[(sp - RTX_LDS_STACK) * spill_stride];
    };

    bool exhausted = false, first_round = true;
    for (;;) {
        // ---------------- refill idle lanes from the queue (ballot + prefix rank) ----------------
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= RTX_REFILL_MIN) {
            uint32_t idx;
            if (first_round) {
                // round one is a static assignment (thread i takes ray i): the queue head was initialised to the grid size
                // by k_begin_batch, so a launch with fewer rays than resident lanes performs no atomic at all
                first_round = false;
                idx = (uint32_t)gtid;
                if ((uint32_t)(gridDim.x * blockDim.x) >= total) exhausted = true;
            } else {
                const int leader = __ffsll((long long)idle) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(head, (uint32_t)n_idle);
                base = __shfl(base, leader);
                if (base + (uint32_t)n_idle >= total) exhausted = true;
                idx = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            }
            if (!has_ray && idx < total) {
                bool live = true;
                if (ANY) {
                    // shadow slot order: light-major within the level, idx = light * lit_n + rank
                    const uint32_t light = idx / lit_n, rank = idx - light * lit_n;
                    out_slot = (uint32_t)q.shadow_base[level] + light * (uint32_t)lit_cap + rank;
                    live = shadow_slot_is_lit(q, level, rank);
                    if (live) { load_shadow_ray(sc, q, level, (int)light, rank, out_slot, wo, wd, tcur); if (tcur < 0.0f) live = false; }
                } else {
                    out_slot = (uint32_t)q.level_base[level] + idx;
                    if (level == 0) {
                        int px, py;
                        live = primary_pixel(sc, q, idx, px, py);
                        RayFull r;
                        if (live) { primary_ray(sc, px, py, r, false); wo = r.o; wd = r.d; }
                    } else {
                        const float4 r0 = q.r0[out_slot], r1 = q.r1[out_slot];
                        wo = V3(r0.x, r0.y, r0.z); wd = V3(r0.w, r1.x, r1.y);
                        live = as_i(r1.z) != RAY_DEAD_PIXEL;          // unused slot of a wave's chunk (fused shading of the level above)
                    }
                    tcur = INFINITY;
                    best_prim = PRIM_NONE; best_tri = 0; best_u = 0.0f; best_v = 0.0f; bvh_steps = 0;
                }
                if (live) {
                    if (COUNT) n_rays++;
                    bool done = false;
                    // Scene::trace_primitives / intersect_primitives: spheres, then planes, then the TLAS (Scene.cpp:173-190)
                    for (int i = 0; i < sc.sphere_count; i++) {
                        if (ANY) { if (sphere_any(sc.spheres[i], wo, wd, tcur)) { done = true; break; } }
                        else { float t; if (sphere_closest(sc.spheres[i], wo, wd, tcur, t)) { tcur = t; best_prim = PRIM_SPHERE | i; } }
                    }
                    if (!done) for (int i = 0; i < sc.plane_count; i++) {
                        const float t = plane_t(sc.planes[i], wo, wd);
                        if (ANY) { if ((t > RAY_EPSILON) & (t < tcur)) { done = true; break; } }
                        else if ((t > RAY_EPSILON) & (t < tcur)) { tcur = t; best_prim = PRIM_PLANE | i; }
                    }
                    if (ANY && done) {
                        q.socc[out_slot] = 1u;
                    } else if (sc.tlas_node_count > 0) {
                        winv = vrcp(wd);
                        co = wo; cd = wd; cinv = winv;
                        cur_nodes = RTX_GPTR(sc.tlas_nodes); floor_sp = -1;
                        sp = 0; push(0);
                        has_ray = true;
                    } else {
                        if (ANY) q.socc[out_slot] = 0u;
                        else { q.h0[out_slot] = make_float4(tcur, 0.0f, 0.0f, as_f(best_prim)); q.h1[out_slot] = 0; }
                    }
                } else if (!ANY) {
                    q.h0[out_slot] = make_float4(INFINITY, 0.0f, 0.0f, as_f(PRIM_NONE));   // clipped pixel: never shaded
                    q.h1[out_slot] = -1;
                }
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---------------- traverse ----------------
        for (int it = 0; it < RTX_TRACE_STEPS; it++) {
            if (!has_ray) continue;
            if (floor_sp >= 0 && sp == floor_sp) {          // BottomLevelBVH::trace returned: back to world space
                floor_sp = -1; co = wo; cd = wd; cinv = winv; cur_nodes = RTX_GPTR(sc.tlas_nodes);
            }
            if (sp == 0) {                                  // ray finished
                if (ANY) q.socc[out_slot] = 0u;
                else { q.h0[out_slot] = make_float4(tcur, best_u, best_v, as_f(best_prim)); q.h1[out_slot] = sc.heatmap ? bvh_steps : best_tri; }
                has_ray = false;
                continue;
            }
            const int e = pop();
            if (e < 0) {
                // Mesh::trace / Mesh::intersect (Mesh.cpp:17-40): ray into model space
                const int inst = sc.tlas_indices[~e];
                const rtx_instance & I = sc.instances[inst];
                co = xform_pos(I.world_inv, wo);
                cd = xform_dir(I.world_inv, wd);
                cinv = vrcp(cd);
                const DevBlas & B = sc.blas[I.blas_id];
                cur_nodes = RTX_GPTR(B.nodes); cur_tris = RTX_GPTR(B.tri_hot); cur_inst = inst;
                floor_sp = sp;
                push(0);
                if (COUNT) n_inst++;
                continue;
            }
            const float4 na = gld(cur_nodes, 2 * e), nb = gld(cur_nodes, 2 * e + 1);
            if (COUNT) { if (floor_sp >= 0) n_blas++; else n_tlas++; }
            if (!slab_test(na, nb, co, cinv, tcur)) continue;
            if (!ANY && floor_sp >= 0) bvh_steps++;         // BottomLevelBVH.cpp:390: BLAS nodes that passed the slab test
            const int first = as_i(na.w), cnt = as_i(nb.w);
            if ((cnt & 0x3fffffff) > 0) {
                if (floor_sp >= 0) {
                    // BLAS leaf: triangle_trace / triangle_intersect (BottomLevelBVH.cpp:214-346)
                    for (int i = first; i < first + cnt; i++) {
                        if (COUNT) n_tri++;
                        const float4 p0 = gld(cur_tris, RTX_TRI_STRIDE * i), e1 = gld(cur_tris, RTX_TRI_STRIDE * i + 1), e2 = gld(cur_tris, RTX_TRI_STRIDE * i + 2);
                        const v3 edge_1 = V3(e1.x, e1.y, e1.z), edge_2 = V3(e2.x, e2.y, e2.z);
                        const v3 h = vcross(cd, edge_2);
                        const float a = vdot(edge_1, h);
                        const float f = 1.0f / a;
                        const v3 s = vsub(co, V3(p0.x, p0.y, p0.z));
                        const float u = f * vdot(s, h);
                        if (!((u > 0.0f) & (u < 1.0f))) continue;
                        const v3 qq = vcross(s, edge_1);
                        const float v = f * vdot(cd, qq);
                        if (!((v > 0.0f) & ((u + v) < 1.0f))) continue;
                        const float t = f * vdot(edge_2, qq);
                        if (!((t > RAY_EPSILON) & (t < tcur))) continue;
                        if (ANY) { q.socc[out_slot] = 1u; has_ray = false; break; }
                        tcur = t; best_u = u; best_v = v; best_tri = i; best_prim = PRIM_TRI | cur_inst;
                    }
                } else {
                    // TLAS leaf: instances are entered in order first .. first+count-1 (TopLevelBVH.cpp:71-75)
                    for (int i = first + cnt - 1; i >= first; i--) push(~i);
                }
            } else {
                // BVHNode::should_visit_left_first, BVHNode.h:30-40
                bool left_first = true;
                if (sc.traversal_strategy == RTX_TRAVERSE_ORDERED) {
                    const int axis = (int)((uint32_t)cnt >> 30);
                    if (axis == 1) left_first = cd.x > 0.0f;
                    else if (axis == 2) left_first = cd.y > 0.0f;
                    else if (axis == 3) left_first = cd.z > 0.0f;
                }
                if (left_first) { push(first + 1); push(first); }
                else            { push(first); push(first + 1); }
            }
        }
    }

    if (n_stackerr) atomicOr(&ctr->error_flags, ERR_STACK_OVERFLOW);
    if (COUNT) {
        // wave reduction, one atomic per counter per wave
        for (int off = 32; off > 0; off >>= 1) {
            n_tlas += __shfl_down(n_tlas, off); n_blas += __shfl_down(n_blas, off);
            n_inst += __shfl_down(n_inst, off); n_tri  += __shfl_down(n_tri, off);
            n_rays += __shfl_down(n_rays, off);
        }
        if (lane == 0) {
            atomicAdd(&ctr->work[ANY ? W_TLAS_A : W_TLAS_C], (unsigned long long)n_tlas);
            atomicAdd(&ctr->work[ANY ? W_BLAS_A : W_BLAS_C], (unsigned long long)n_blas);
            atomicAdd(&ctr->work[ANY ? W_INST_A : W_INST_C], (unsigned long long)n_inst);
            atomicAdd(&ctr->work[ANY ? W_TRI_A  : W_TRI_C],  (unsigned long long)n_tri);
            atomicAdd(&ctr->work[ANY ? W_ANY_RAYS : W_CLOSEST_RAYS], (unsigned long long)n_rays);
        }
    }
}


// =================================================================================================
// k_trace_fast — same rays, same per-lane traversal ORDER and therefore the same hits as k_trace,
// with two changes that cut the number of dependent memory round trips per ray roughly in half:
//
//  * child PAIR fetch: the two children of an inner node are adjacent (BVHNode.h:10-16,
//    BVHBuilders.h:20-21), so one 64-byte fetch (4 x dwordx4) brings both boxes; both are slab-tested
//    at once.
//  * early filter: a child that fails the slab test now is not pushed.  This is exact: the reference
//    tests a node when it POPS it, against a closest distance that can only have shrunk since, and
//    `t_near < min(min(D, tx), min(ty, tz))` is monotone in D — failing now implies failing later.
//    The near child is processed immediately (the reference pops it next, with D unchanged); the far
//    child is pushed and RE-TESTED when popped, against the then-current D, which is exactly the
//    reference's pop-time test.  For shadow rays D never changes, so the re-test is skipped.
//
// Work counters are produced by k_trace<ANY, true> (reference-order accounting); this kernel has none.
template <bool ANY>
__global__ __launch_bounds__(RTX_TRACE_BLOCK, ANY ? RTX_WAVES_ANY : RTX_WAVES_CLOSEST)
void k_trace_fast(const DevScene sc, const DevQueues q, const int level, const int level_hi) {
    __shared__ int lds_stack[RTX_TRACE_BLOCK / RTX_WAVE][RTX_LDS_STACK][RTX_WAVE];
    // closest-hit rays carry a second word per entry: the pop-time key of slab_test_key (shadow rays never re-test)
    __shared__ float lds_key[ANY ? 1 : RTX_TRACE_BLOCK / RTX_WAVE][ANY ? 1 : RTX_LDS_STACK][RTX_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int * const stk = &lds_stack[wave][0][lane];
    float * const stk_key = &lds_key[ANY ? 0 : wave][0][lane];
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    // spill regions (ints): [0, S) entries of closest-hit kernels, [S, 2S) their keys, [2S, 3S) shadow-ray kernels, S = threads * depth:
    // a closest-hit and a shadow-ray launch may be resident together (two streams), so they must not share a region
    const int spill_stride = q.spill_threads;
    const size_t spill_region = (size_t)spill_stride * (RTX_MAX_STACK - RTX_LDS_STACK);
    int * const spill = q.spill + (ANY ? 2 * spill_region : 0) + gtid;
    float * const spill_key = (float *)(q.spill + spill_region) + gtid;

    DevCounters * const ctr = q.counters;
    // shadow rays: one launch may cover the levels [level, level_hi]; closest-hit rays: exactly one level
    uint32_t total = 0;
    if (ANY) { for (int d = level; d <= level_hi; d++) total += shadow_slots(ctr, q, d) * (uint32_t)sc.light_count; }
    else total = ctr->ray_count[level];
    uint32_t * const head = ANY ? &ctr->fetch_any[level] : &ctr->fetch_closest[level];
    const bool ordered = sc.traversal_strategy == RTX_TRAVERSE_ORDERED;

    bool has_ray = false, have_cur = false, lane_risky = false, risky_world = false;   // lane_risky: a slab test of this ray in the space it is traversing can produce a NaN (pk_nan_possible)
    uint32_t out_slot = 0;
    v3 wo = V3(0, 0, 0), wd = wo, winv = wo, co = wo, cd = wo, cinv = wo;
    float tcur = 0.0f;
    int best_prim = PRIM_NONE, best_tri = 0; float best_u = 0.0f, best_v = 0.0f;
    int sp = 0, floor_sp = -1, cur_inst = 0;
    int cur_first = 0, cur_cnt = 0;                 // the node being visited: it has already passed its slab test
    rtx_gptr cur_nodes = RTX_GPTR(sc.tlas_nodes);
    rtx_gptr cur_tris  = nullptr;
    uint32_t n_stackerr = 0;

    auto push = [&](int e, float key) {
        if (sp < RTX_LDS_STACK) { stk[sp * RTX_WAVE] = e; if (!ANY) stk_key[sp * RTX_WAVE] = key; }
        else if (sp < RTX_MAX_STACK) { spill[(sp - RTX_LDS_STACK) * spill_stride] = e; if (!ANY) spill_key[(sp - RTX_LDS_STACK) * spill_stride] = key; }
        else { n_stackerr = 1; return; }
        sp++;
    };
    auto pop = [&](float & key) -> int {
        sp--;
        if (sp < RTX_LDS_STACK) { if (!ANY) key = stk_key[sp * RTX_WAVE]; return stk[sp * RTX_WAVE]; }
        if (!ANY) key = spill_key[(sp - RTX_LDS_STACK) * spill_stride];
        return spill[(sp - RTX_LDS_STACK) * spill_stride];
    };
    auto finish = [&]() {
        if (ANY) q.socc[out_slot] = 0u;
        else { q.h0[out_slot] = make_float4(tcur, best_u, best_v, as_f(best_prim)); q.h1[out_slot] = best_tri; }
        has_ray = false; have_cur = false;
    };

    bool exhausted = false, first_round = true;
    for (;;) {
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= RTX_REFILL_MIN) {
            uint32_t idx;
            if (first_round) {
                // round one is a static assignment (thread i takes ray i): the queue head was initialised to the grid size
                // by k_begin_batch, so a launch with fewer rays than resident lanes performs no atomic at all
                first_round = false;
                idx = (uint32_t)gtid;
                if ((uint32_t)(gridDim.x * blockDim.x) >= total) exhausted = true;
            } else {
                const int leader = __ffsll((long long)idle) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(head, (uint32_t)n_idle);
                base = __shfl(base, leader);
                if (base + (uint32_t)n_idle >= total) exhausted = true;
                idx = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            }
            if (!has_ray && idx < total) {
                bool live = true;
                if (ANY) {
                    // idx -> (level, light, rank): levels are concatenated, light-major inside a level
                    uint32_t local = idx; int d = level;
                    uint32_t lit_n = shadow_slots(ctr, q, d);
                    while (d < level_hi && local >= lit_n * (uint32_t)sc.light_count) { local -= lit_n * (uint32_t)sc.light_count; d++; lit_n = shadow_slots(ctr, q, d); }
                    const uint32_t light = local / lit_n, rank = local - light * lit_n;
                    out_slot = (uint32_t)q.shadow_base[d] + light * (uint32_t)q.level_cap[d] + rank;
                    live = shadow_slot_is_lit(q, d, rank);
                    if (live) { load_shadow_ray(sc, q, d, (int)light, rank, out_slot, wo, wd, tcur); if (tcur < 0.0f) live = false; }      // negative: culled by k_shade<.., CULL> (contribution exactly zero)
                } else {
                    out_slot = (uint32_t)q.level_base[level] + idx;
                    if (level == 0) {
                        int px, py;
                        live = primary_pixel(sc, q, idx, px, py);
                        RayFull r;
                        if (live) { primary_ray(sc, px, py, r, false); wo = r.o; wd = r.d; }
                    } else {
                        const float4 r0 = q.r0[out_slot], r1 = q.r1[out_slot];
                        wo = V3(r0.x, r0.y, r0.z); wd = V3(r0.w, r1.x, r1.y);
                        live = as_i(r1.z) != RAY_DEAD_PIXEL;
                    }
                    tcur = INFINITY;
                    best_prim = PRIM_NONE; best_tri = 0; best_u = 0.0f; best_v = 0.0f;
                }
                if (live) {
                    bool done = false;
                    for (int i = 0; i < sc.sphere_count; i++) {
                        if (ANY) { if (sphere_any(sc.spheres[i], wo, wd, tcur)) { done = true; break; } }
                        else { float t; if (sphere_closest(sc.spheres[i], wo, wd, tcur, t)) { tcur = t; best_prim = PRIM_SPHERE | i; } }
                    }
                    if (!done) for (int i = 0; i < sc.plane_count; i++) {
                        const float t = plane_t(sc.planes[i], wo, wd);
                        if (ANY) { if ((t > RAY_EPSILON) & (t < tcur)) { done = true; break; } }
                        else if ((t > RAY_EPSILON) & (t < tcur)) { tcur = t; best_prim = PRIM_PLANE | i; }
                    }
                    if (ANY && done) {
                        q.socc[out_slot] = 1u;
                    } else if (sc.tlas_node_count > 0) {
                        winv = vrcp(wd);
                        co = wo; cd = wd; cinv = winv; risky_world = !inv_is_finite(winv) && pk_nan_possible_tlas(sc, wo, winv); lane_risky = risky_world;
                        cur_nodes = RTX_GPTR(sc.tlas_nodes); floor_sp = -1; sp = 0;
                        has_ray = true;
                        // TLAS root: pushed, popped and tested by the reference (TopLevelBVH.cpp:58-67)
                        const float4 na = gld(cur_nodes, 0), nb = gld(cur_nodes, 1);
                        if (slab_test(na, nb, co, cinv, tcur)) { cur_first = as_i(na.w); cur_cnt = as_i(nb.w); have_cur = true; }
                        else finish();
                    } else {
                        if (ANY) q.socc[out_slot] = 0u;
                        else { q.h0[out_slot] = make_float4(tcur, 0.0f, 0.0f, as_f(best_prim)); q.h1[out_slot] = 0; }
                    }
                } else if (!ANY) {
                    q.h0[out_slot] = make_float4(INFINITY, 0.0f, 0.0f, as_f(PRIM_NONE));
                    q.h1[out_slot] = -1;
                }
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) break;
            continue;
        }

        for (int it = 0; it < RTX_TRACE_STEPS; it++) {
            // all live rays of the wave free of +-0 direction components in the space they are traversing?  (lane_risky follows cinv)
            const bool wave_nan_free = __ballot(has_ray && lane_risky) == 0ull;
            // ---- A: inner node: fetch the child pair, test both, descend into the near one ----
            // (repeated while most lanes are still walking inner nodes: lanes that reached a leaf or ran dry wait a little,
            //  which keeps the wave converged on the dominant step)
            #pragma unroll 1
            for (int rep = 0; rep < RTX_INNER_REPEAT; rep++) {
            if (rep > 0 && __popcll(__ballot(has_ray && have_cur && (cur_cnt & 0x3fffffff) == 0)) < RTX_INNER_MIN_LANES) break;
            if (has_ray && have_cur && (cur_cnt & 0x3fffffff) == 0) {
                const int left = cur_first;
                const float4 la = gld(cur_nodes, 2 * left), lb = gld(cur_nodes, 2 * left + 1), ra = gld(cur_nodes, 2 * left + 2), rb = gld(cur_nodes, 2 * left + 3);
                float kl = 0.0f, kr = 0.0f;
                bool pl, pr;
                if ((ANY ? RTX_NANFREE_ANY : RTX_NANFREE_CLOSEST) && wave_nan_free) { pl = slab_test_sel<true>(la, lb, co, cinv, tcur, kl);  pr = slab_test_sel<true>(ra, rb, co, cinv, tcur, kr); }
                else               { pl = slab_test_sel<false>(la, lb, co, cinv, tcur, kl); pr = slab_test_sel<false>(ra, rb, co, cinv, tcur, kr); }
                bool left_first = true;                                  // BVHNode::should_visit_left_first, BVHNode.h:30-40
                if (ordered) {
                    const int axis = (int)((uint32_t)cur_cnt >> 30);
                    if (axis == 1) left_first = cd.x > 0.0f;
                    else if (axis == 2) left_first = cd.y > 0.0f;
                    else if (axis == 3) left_first = cd.z > 0.0f;
                }
                const bool p_near = left_first ? pl : pr, p_far = left_first ? pr : pl;
                const int near_first = left_first ? as_i(la.w) : as_i(ra.w), near_cnt = left_first ? as_i(lb.w) : as_i(rb.w);
                const int far_first  = left_first ? as_i(ra.w) : as_i(la.w), far_cnt  = left_first ? as_i(rb.w) : as_i(lb.w);
                if (p_near) {
                    if (p_far) {
                        // the far child's (first, count) travel on the stack, so the pop needs no memory access: shadow rays never
                        // re-test it (their max distance is constant); closest-hit rays re-test it with its key (slab_test_key)
                        const int far_leaf = far_cnt & 0x3fffffff;
                        const float far_key = left_first ? kr : kl;
                        if ((uint32_t)far_first < (1u << 24) && far_leaf < 16)
                            push((int)(0x40000000u | (((uint32_t)far_cnt >> 30) << 28) | ((uint32_t)far_leaf << 24) | (uint32_t)far_first), far_key);
                        else push(left_first ? left + 1 : left, far_key);
                    }
                    cur_first = near_first; cur_cnt = near_cnt;
                } else if (p_far) {
                    cur_first = far_first; cur_cnt = far_cnt;
                } else have_cur = false;
            }
            }
            // ---- B: leaf ----
            if (has_ray && have_cur && (cur_cnt & 0x3fffffff) > 0) {
                have_cur = false;
                const int first = cur_first, cnt = cur_cnt;
                if (floor_sp >= 0) {
                    for (int i = first; i < first + cnt; i++) {
                        const float4 p0 = gld(cur_tris, RTX_TRI_STRIDE * i), e1 = gld(cur_tris, RTX_TRI_STRIDE * i + 1), e2 = gld(cur_tris, RTX_TRI_STRIDE * i + 2);
                        const v3 edge_1 = V3(e1.x, e1.y, e1.z), edge_2 = V3(e2.x, e2.y, e2.z);
                        const v3 h = vcross(cd, edge_2);
                        const float a = vdot(edge_1, h);
                        const float f = 1.0f / a;
                        const v3 s = vsub(co, V3(p0.x, p0.y, p0.z));
                        const float u = f * vdot(s, h);
                        if (!((u > 0.0f) & (u < 1.0f))) continue;
                        const v3 qq = vcross(s, edge_1);
                        const float v = f * vdot(cd, qq);
                        if (!((v > 0.0f) & ((u + v) < 1.0f))) continue;
                        const float t = f * vdot(edge_2, qq);
                        if (!((t > RAY_EPSILON) & (t < tcur))) continue;
                        if (ANY) { q.socc[out_slot] = 1u; has_ray = false; break; }
                        tcur = t; best_u = u; best_v = v; best_tri = i; best_prim = PRIM_TRI | cur_inst;
                    }
                } else {
                    for (int i = first + cnt - 1; i >= first; i--) push(~i, 0.0f);
                }
            }
            // ---- C: pop ----
            if (has_ray && !have_cur) {
                if (floor_sp >= 0 && sp == floor_sp) { floor_sp = -1; co = wo; cd = wd; cinv = winv; cur_nodes = RTX_GPTR(sc.tlas_nodes); lane_risky = risky_world; }
                if (sp == 0) { finish(); continue; }
                float key = 0.0f;
                const int e = pop(key);
                if (e < 0) {
                    const int inst = sc.tlas_indices[~e];                // Mesh::trace / intersect, Mesh.cpp:17-40
                    const rtx_instance & I = sc.instances[inst];
                    co = xform_pos(I.world_inv, wo);
                    cd = xform_dir(I.world_inv, wd);
                    cinv = vrcp(cd);
                    const DevBlas & B = sc.blas[I.blas_id];
                    lane_risky = !inv_is_finite(cinv) && pk_nan_possible(B, co, cinv);
                    cur_nodes = RTX_GPTR(B.nodes); cur_tris = RTX_GPTR(B.tri_hot); cur_inst = inst;
                    floor_sp = sp;
                    const float4 na = gld(cur_nodes, 0), nb = gld(cur_nodes, 1);    // BLAS root
                    if (slab_test(na, nb, co, cinv, tcur)) { cur_first = as_i(na.w); cur_cnt = as_i(nb.w); have_cur = true; }
                } else if (ANY || key < tcur) {
                    // far child: the reference tests it now, against the current closest distance == `key < tcur` (slab_test_key)
                    if (e & 0x40000000) {
                        cur_first = e & 0x00ffffff;
                        cur_cnt = (int)((((uint32_t)e >> 28) & 3u) << 30) | ((e >> 24) & 15);
                    } else {
                        const float4 na = gld(cur_nodes, 2 * e), nb = gld(cur_nodes, 2 * e + 1);   // too big to pack: read (first, count) back
                        cur_first = as_i(na.w); cur_cnt = as_i(nb.w);
                    }
                    have_cur = true;
                }
            }
        }
    }
    if (n_stackerr) atomicOr(&ctr->error_flags, ERR_STACK_OVERFLOW);
}
