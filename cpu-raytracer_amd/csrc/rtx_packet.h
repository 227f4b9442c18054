// rtx_packet.h — wave-packet traversal: the production closest-hit and any-hit kernels.
//
// A wavefront walks the BVH for the 64 rays of one PACKET together (the reference's own idea — Raytracer::render_tile traces
// SIMD_LANE_SIZE-wide packets, BottomLevelBVH.cpp:355-437 — at the width of a CDNA4 wave):
//
//   * ONE traversal stack per wave holding (node, 64-bit lane mask).  A lane is in a node's mask iff every box on the chain from
//     the root passed THAT lane's own slab test — exactly the set of nodes the reference's per-ray loop would pop and pass for
//     that ray (SIMD_LANE_SIZE 1), so each lane tests exactly the triangles the reference tests for it.  The stack lives in six
//     VGPRs (lane i = entry i, v_writelane / v_readlane): no LDS, no memory latency on push / pop.
//   * node pairs and triangles are wave-uniform: they are fetched with SCALAR loads (s_load_dwordx8/x16 through the scalar data
//     cache) into SGPRs and fed to the VALU as scalar operands.  No per-lane gather, no per-lane stack: the resource the per-lane
//     kernels of rtx_trace.h are bound by (the CU's divergent-gather rate) is not used at all.
//   * masks are v_cmp results combined with SALU; control flow is wave-uniform (s_cbranch), there is no divergence.
//   * the slab test is written for the packed-fp32 VALU (v_pk_add_f32 / v_pk_mul_f32): the device node record is
//     (min.x, min.y, max.x, max.y) (min.z, max.z, first, count), so each SGPR pair is one packed operand.
//
// Exactness (the parity contract is bit-identical hits):
//   any-hit      the result "some triangle in a leaf this ray reaches is hit" does not depend on visiting order; per-lane masks keep
//                the reached-leaf set identical, so occlusion flags equal the reference's.
//   closest-hit  the reference's result depends on ORDER (equal-t ties: first found wins; culling against the closest distance so
//                far).  Lanes are independent, and a lane's own order is preserved as long as every inner node it visits is
//                expanded in that lane's near/far order.  BVHNode::should_visit_left_first (BVHNode.h:30-40) depends only on the
//                sign of one direction component, so at a node where the lanes of the mask disagree the mask is SPLIT: the lanes
//                that want the other order are parked on the stack as a "pretested" entry for the same node and walk its subtree
//                afterwards.  The far child is pushed with the lanes that pass now and re-tested at pop time against each lane's
//                then-current closest distance — the reference's pop-time test (slab test monotone in the distance: a lane that
//                fails now would fail later).
//   NaN          min/max are `a<b?a:b` in the reference (NaN -> b).  A slab test can only produce a NaN as 0 * inf, i.e. when an
//                inverse direction component is infinite; packets whose lanes all have finite inverse directions run the FAST
//                variant (v_min/v_max/v_min3/v_max3, identical on non-NaN operands), all others the reference-form variant.
//   stack size   the reference's per-BVH stack (BVH_TRAVERSAL_STACK_SIZE, Config.h:25) holds one pending far sibling per ancestor
//                entered through its near child and overflows when an inner node is expanded with size - 1 or more pending.  That
//                count is wave-uniform here (`depth`), travels with the stack entries, and raises ERR_STACK_OVERFLOW at exactly the
//                configured rtx_config.stack_size.
#pragma once
#include "rtx_math.h"
#include "rtx_device.h"
#include "rtx_trace.h"
#include "rtx_shade.h"          // shade_ray: the closest-hit packet kernel shades its own hits (FUSE)

#ifndef RTX_PK_BLOCK
#define RTX_PK_BLOCK    64           // the waves of a packet launch are independent: one wave per workgroup finds room beside other frames' kernels
                                     // soonest (with k_items at 64 too: 1.318 -> 1.275 ms per frame; either one alone: no change)
#endif
#define RTX_PK_STACK    64           // packet stack entries per wave = lanes of the stack VGPRs
#define RTX_PK_CLASSES  8            // packet-queue heads per launch (packet k belongs to head k % 8): spreads the fetch atomics over 8 addresses
#ifndef RTX_PK_WAVES_ANY
#define RTX_PK_WAVES_ANY 5           // __launch_bounds__ min waves per SIMD (93 VGPRs unconstrained; forcing 8 waves spills 26 dwords per lane)
#endif
#ifndef RTX_PK_WAVES_SPLIT
#define RTX_PK_WAVES_SPLIT 5         // the split shadow-ray kernel (no per-lane phases): 78 VGPRs, no spills
#endif
#ifndef RTX_PK_WAVES_CLOSEST
#define RTX_PK_WAVES_CLOSEST 4
#endif
#ifndef RTX_PK_CHUNK
#define RTX_PK_CHUNK 64u             // FUSE: child-ray slots a wave takes from the next level's queue per atomic (whole 64-slot packets; the unused tail of a wave's last chunk stays empty)
#endif
#ifndef RTX_PK_WAVES_FUSED
#define RTX_PK_WAVES_FUSED 3         // closest-hit walk + shading of the hits in one kernel: shade_ray needs 168 VGPRs; the closest-hit walk does not mind
                                     // (compiled for 3 / 4 / 5 waves per SIMD it takes 0.828 / 0.839 / 0.934 ms per frame alone)
#endif

// constant address space + wave-uniform index = scalar load
typedef const __attribute__((address_space(4))) rtx_f4v * rtx_cptr;
typedef float rtx_f2v __attribute__((ext_vector_type(2)));
RTX_D int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
RTX_D unsigned long long rfl64(unsigned long long v) { return ((unsigned long long)(uint32_t)rfl((int)(uint32_t)(v >> 32)) << 32) | (unsigned long long)(uint32_t)rfl((int)(uint32_t)v); }
RTX_D rtx_cptr pk_cptr(const void * p) {          // pointers out of device tables arrive in VGPRs: make them wave-uniform (SGPR pair)
    const unsigned long long a = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)(uint32_t)rfl((int)(uint32_t)(a >> 32)) << 32) | (unsigned long long)(uint32_t)rfl((int)(uint32_t)a);
    return (rtx_cptr)u;
}
RTX_D float4 sld(rtx_cptr p, int i) { const rtx_f4v v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }

// v_min / v_max without the canonicalisation clang adds for fminf / fmaxf (operands here are never NaN: FAST packets only)
RTX_D float pk_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTX_D float pk_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTX_D float pk_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RTX_D float pk_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RTX_D float pk_max_s(float s, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(b)); return r; }

#define PK_FLAG_ITER      0x40000000      // df: entry iterates over the instances of a TLAS leaf (a = next leaf slot, b = remaining)
#define PK_FLAG_PRETESTED 0x20000000      // df: lanes of the mask already passed this node's slab test (sign split)

// packet-walk statistics (RTX_RENDER_PACKET_STATS), rtx_work_counters::reserved order
enum { PKS_PACKETS, PKS_PAIR_STEPS, PKS_TRI_STEPS, PKS_PAIR_LANES, PKS_TRI_LANES };

struct PkRay {                       // per lane: the ray in the space being walked, paired for the packed-fp32 VALU
    rtx_f2v o_xy, o_zz, i_xy, i_zz;  // origin (x, y) (z, z), inverse direction (x, y) (z, z)
    v3 d;
    float tcur;                      // closest distance so far (any-hit: the constant maximum distance)
};
RTX_D void pk_set_ray(PkRay & r, v3 o, v3 d, v3 inv) {
    r.o_xy = rtx_f2v{ o.x, o.y }; r.o_zz = rtx_f2v{ o.z, o.z };
    r.i_xy = rtx_f2v{ inv.x, inv.y }; r.i_zz = rtx_f2v{ inv.z, inv.z };
    r.d = d;
}

// AABB::intersect (AABB.cpp:38-52) on a packet node record q0 = (min.x, min.y, max.x, max.y), q1 = (min.z, max.z, -, -)
template <bool FAST>
RTX_D bool pk_slab(float4 q0, float4 q1, const PkRay & r) {
    const rtx_f2v t0xy = (rtx_f2v{ q0.x, q0.y } - r.o_xy) * r.i_xy;      // (t0.x, t0.y)
    const rtx_f2v t1xy = (rtx_f2v{ q0.z, q0.w } - r.o_xy) * r.i_xy;      // (t1.x, t1.y)
    const rtx_f2v tz   = (rtx_f2v{ q1.x, q1.y } - r.o_zz) * r.i_zz;      // (t0.z, t1.z)
    if (FAST) {
        const float t_near = pk_max_s(RAY_EPSILON, pk_max3(pk_min(t0xy.x, t1xy.x), pk_min(t0xy.y, t1xy.y), pk_min(tz.x, tz.y)));
        const float t_far  = pk_min(r.tcur, pk_min3(pk_max(t0xy.x, t1xy.x), pk_max(t0xy.y, t1xy.y), pk_max(tz.x, tz.y)));
        return t_near < t_far;
    }
    const float tminx = fmin_ref(t0xy.x, t1xy.x), tminy = fmin_ref(t0xy.y, t1xy.y), tminz = fmin_ref(tz.x, tz.y);
    const float tmaxx = fmax_ref(t0xy.x, t1xy.x), tmaxy = fmax_ref(t0xy.y, t1xy.y), tmaxz = fmax_ref(tz.x, tz.y);
    const float t_near = fmax_ref(fmax_ref(RAY_EPSILON, tminx), fmax_ref(tminy, tminz));
    const float t_far  = fmin_ref(fmin_ref(r.tcur, tmaxx), fmin_ref(tmaxy, tmaxz));
    return t_near < t_far;
}

#ifdef RTX_PK_PROF
#define PK_T0() const unsigned long long pk_t0_ = __builtin_amdgcn_s_memtime()
#define PK_T1(slot) do { W.prof[slot] += __builtin_amdgcn_s_memtime() - pk_t0_; W.prof[slot + 5]++; } while (0)
#else
#define PK_T0() do {} while (0)
#define PK_T1(slot) do {} while (0)
#endif
struct PkWalk {                      // everything a packet walk carries; wave-uniform unless noted
    rtx_cptr tlas, nodes, tris;
    rtx_gptr lane_nodes, lane_nodes4c, lane_tris;                      // the BLAS being walked in the per-lane kernels' layout (per-lane phases)
    int sp, floor_sp, cur_inst, depth;
    int cur_idx, cur_first, cur_cnt;
    unsigned long long m, alive;
    int st_a, st_b, st_mlo, st_mhi, st_idx, st_df;      // per lane: the stack, lane i = entry i
    PkRay ray;                                           // per lane
    bool valid, occluded;                                // per lane
    int best_prim, best_tri; float best_u, best_v;       // per lane (closest-hit)
    unsigned long long pub;                              // closest-hit hybrid: lanes still walking with the packet (the others turned private)
    int fifo_n;                                          // per lane: entries on the lane's private work list (closest-hit hybrid)
    int psp;                                             // per lane: entries on the lane's PRIVATE stack (hybrid any-hit walk)
    int pk_steps;                                        // steps of the current packet walk (the deferral threshold grows with it)
    int item_n, cur_blas; uint32_t item_slot;            // split walk: items this wave has appended to its chunk, the mesh being walked; per lane: the ray's slot
    uint32_t err;
    unsigned long long st_pairs, st_tris, st_pair_lanes, st_tri_lanes;
#ifdef RTX_PK_PROF
    unsigned long long prof[10];      // cycles: 0 setup, 1 descend, 2 leaf, 3 pop, 4 results; 5..9 their counts
#endif
#ifdef RTX_LANE_PROF
    // closest-hit per-lane phase of ONE level (RTX_LANE_PROF_LEVEL): where do the dependent steps' microseconds go?  (tools/lane_prof.py)
    uint32_t lp[16];                  // (kept wave-uniform in SGPRs: every update goes through readfirstlane) 0 phase ticks, 1 iterations, 2 inner steps, 3 leaf steps, 4 pops with a node fetch, 5 work-list reads, 6 lanes still walking summed over the iterations,
                                      // 7 ticks waiting for child pairs, 8 for triangles, 9 for pop / work-list fetches, 10 max phase ticks, 11 max iterations, 12 phases
#endif
};

// v_writelane_b32: hipcc declares no __builtin_amdgcn_writelane, so the LLVM intrinsic is bound by name
extern "C" __device__ int rtx_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// Push: the (wave-uniform) fields go into lane `sp` of the stack VGPRs
template <bool ANY>
RTX_D void pk_push(PkWalk & W, int a, int b, unsigned long long mask, int idx, int df) {
    if (W.sp >= RTX_PK_STACK) { W.err = 1; return; }
    W.st_a   = rtx_writelane(a, W.sp, W.st_a);
    W.st_b   = rtx_writelane(b, W.sp, W.st_b);
    W.st_mlo = rtx_writelane((int)(uint32_t)mask, W.sp, W.st_mlo);
    W.st_mhi = rtx_writelane((int)(uint32_t)(mask >> 32), W.sp, W.st_mhi);
    W.st_df  = rtx_writelane(df, W.sp, W.st_df);
    if (!ANY) W.st_idx = rtx_writelane(idx, W.sp, W.st_idx);      // node index: only the closest-hit pop-time re-test needs it
    W.sp++;
}

// ---- the descend loop, hand-scheduled --------------------------------------------------------------------------------------
// From an inner node (cur_first, cur_cnt, m != 0) walk down — child-pair fetch, two slab tests, near / far choice, push of the far
// child — until a leaf is reached (returns with the leaf in cur_* and its mask in m) or no lane is left (m == 0).  This is the inner
// loop of the FAST walk (finite inverse directions: v_min / v_max semantics equal the reference's on non-NaN operands) written as
// one asm block, because the compiler keeps the wave-uniform state of this loop in VGPRs (selects through v_cndmask, address
// arithmetic on the VALU, copies of the stack registers at every join): 124 VALU instructions per step compiled vs 34 + 5 here.
//   SGPR temporaries s[64:95] (child pair in s[64:79]), VGPR temporaries v[24:31]; both are declared as clobbers.
//   Hazards (gfx940/950 table): every VALU -> VALU dependency is through VGPRs (interlocked); VCC written by v_cmp is read by SALU
//   only (interlocked); M0 is written by SALU before v_writelane (no wait state required, one s_nop kept); no trans ops, no DPP.
#define PK_ASM_SLAB(X0, X1, X2, X3, X4, X5, MASK)                                                                                   \
    "v_pk_add_f32 v[24:25], s[" #X0 ":" #X1 "], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_add_f32 v[26:27], s[" #X2 ":" #X3 "], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_add_f32 v[28:29], s[" #X4 ":" #X5 "], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_mul_f32 v[24:25], %[ixy], v[24:25]\n\t"                                                                                   \
    "v_pk_mul_f32 v[26:27], %[ixy], v[26:27]\n\t"                                                                                   \
    "v_pk_mul_f32 v[28:29], %[izz], v[28:29]\n\t"                                                                                   \
    "v_min_f32 v30, v24, v26\n\t"                                                                                                   \
    "v_max_f32 v24, v24, v26\n\t"                                                                                                   \
    "v_min_f32 v31, v25, v27\n\t"                                                                                                   \
    "v_max_f32 v25, v25, v27\n\t"                                                                                                   \
    "v_min_f32 v26, v28, v29\n\t"                                                                                                   \
    "v_max_f32 v27, v28, v29\n\t"                                                                                                   \
    "v_max3_f32 v30, v30, v31, v26\n\t"                                                                                             \
    "v_min3_f32 v24, v24, v25, v27\n\t"                                                                                             \
    "v_max_f32 v30, %[eps], v30\n\t"                                                                                                \
    "v_min_f32 v24, %[tcur], v24\n\t"                                                                                               \
    "v_cmp_lt_f32 vcc, v30, v24\n\t"                                                                                                \
    "s_and_b64 " MASK ", vcc, %[m]\n\t"

// The same test for a packet whose lanes agree on the sign of every inverse-direction component: with box min <= max, (b - o) * inv is monotone
// in b, so the smaller of the two plane distances of an axis is known per PACKET (t0 where inv >= 0, t1 where inv < 0) and the six
// v_min / v_max disappear: 11 VALU per box instead of 17.  NX .. FZ name the registers holding the near / far distance of each axis
// (v24 = t0.x, v25 = t0.y, v26 = t1.x, v27 = t1.y, v28 = t0.z, v29 = t1.z).
#define PK_ASM_SLAB_S(X0, X1, X2, X3, X4, X5, MASK, NX, FX, NY, FY, NZ, FZ)                                                          \
    "v_pk_add_f32 v[24:25], s[" #X0 ":" #X1 "], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_add_f32 v[26:27], s[" #X2 ":" #X3 "], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_add_f32 v[28:29], s[" #X4 ":" #X5 "], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                              \
    "v_pk_mul_f32 v[24:25], %[ixy], v[24:25]\n\t"                                                                                   \
    "v_pk_mul_f32 v[26:27], %[ixy], v[26:27]\n\t"                                                                                   \
    "v_pk_mul_f32 v[28:29], %[izz], v[28:29]\n\t"                                                                                   \
    "v_max3_f32 v30, " NX ", " NY ", " NZ "\n\t"                                                                                    \
    "v_min3_f32 v31, " FX ", " FY ", " FZ "\n\t"                                                                                    \
    "v_max_f32 v30, %[eps], v30\n\t"                                                                                                \
    "v_min_f32 v31, %[tcur], v31\n\t"                                                                                               \
    "v_cmp_lt_f32 vcc, v30, v31\n\t"                                                                                                \
    "s_and_b64 " MASK ", vcc, %[m]\n\t"
#define PK_ASM_QUAD_S(NX, FX, NY, FY, NZ, FZ)                                                                                       \
    PK_ASM_SLAB_S(48, 49, 50, 51, 52, 53, "s[80:81]", NX, FX, NY, FY, NZ, FZ)                                                       \
    PK_ASM_SLAB_S(56, 57, 58, 59, 60, 61, "s[82:83]", NX, FX, NY, FY, NZ, FZ)                                                       \
    PK_ASM_SLAB_S(64, 65, 66, 67, 68, 69, "s[84:85]", NX, FX, NY, FY, NZ, FZ)                                                       \
    PK_ASM_SLAB_S(72, 73, 74, 75, 76, 77, "s[86:87]", NX, FX, NY, FY, NZ, FZ)                                                       \
    "s_branch PKB_SEL_%=\n\t"

template <bool ANY>
RTX_D void pk_descend(PkWalk & W, int stack_size, int ordered, int thr) {
    // "s" operands must really be SGPRs: whatever the compiler did with the wave-uniform state, pin it here
    W.cur_first = rfl(W.cur_first); W.cur_cnt = rfl(W.cur_cnt); W.cur_idx = rfl(W.cur_idx); W.m = rfl64(W.m); W.sp = rfl(W.sp);
    W.depth = rfl(W.depth); W.err = (uint32_t)rfl((int)W.err); stack_size = rfl(stack_size); ordered = rfl(ordered); thr = rfl(thr);
    int steps = rfl(W.pk_steps);
    const unsigned long long posx = __ballot(W.ray.d.x > 0.0f), posy = __ballot(W.ray.d.y > 0.0f), posz = __ballot(W.ray.d.z > 0.0f);
    const float eps = RAY_EPSILON;
    const unsigned long long nodes = rfl64((unsigned long long)W.nodes);
    if (ANY) {
        asm volatile(
            "s_mov_b64 s[94:95], %[nodes]\n\t"
            "PKA_TOP_%=:\n\t"
            // reference stack emulation: `depth` pending + 2 pushed
            "s_add_i32 s90, %[depth], 2\n\t"
            "s_cmp_gt_i32 s90, %[ssz]\n\t"
            "s_cselect_b32 %[err], 1, %[err]\n\t"
            // child pair: 64 bytes at nodes + first * 32
            "s_mov_b32 s92, %[first]\n\t"
            "s_mov_b32 s93, 0\n\t"
            "s_lshl_b64 s[92:93], s[92:93], 5\n\t"
            "s_add_u32 s92, s92, s94\n\t"
            "s_addc_u32 s93, s93, s95\n\t"
            "s_load_dwordx16 s[64:79], s[92:93], 0x0\n\t"
            // order (any-hit: any order is exact; the first lane's): left_first = !ordered | axis == 0 | pos_axis[first lane of m]
            "s_lshr_b32 s90, %[cnt], 30\n\t"
            "s_mov_b64 s[86:87], %[posx]\n\t"
            "s_cmp_eq_u32 s90, 2\n\t"
            "s_cselect_b64 s[86:87], %[posy], s[86:87]\n\t"
            "s_cmp_eq_u32 s90, 3\n\t"
            "s_cselect_b64 s[86:87], %[posz], s[86:87]\n\t"
            "s_ff1_i32_b64 s91, %[m]\n\t"
            "s_bitcmp1_b64 s[86:87], s91\n\t"
            "s_cselect_b32 s91, 1, 0\n\t"
            "s_cmp_eq_u32 s90, 0\n\t"
            "s_cselect_b32 s91, 1, s91\n\t"
            "s_cmp_eq_u32 %[ordered], 0\n\t"
            "s_cselect_b32 s91, 1, s91\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[82:83]")
            PK_ASM_SLAB(72, 73, 74, 75, 76, 77, "s[84:85]")
            // near / far
            "s_cmp_lg_u32 s91, 0\n\t"
            "s_cselect_b64 s[86:87], s[82:83], s[84:85]\n\t"
            "s_cselect_b64 s[88:89], s[84:85], s[82:83]\n\t"
            "s_cselect_b32 s80, s70, s78\n\t"
            "s_cselect_b32 s81, s71, s79\n\t"
            "s_cselect_b32 s92, s78, s70\n\t"
            "s_cselect_b32 s93, s79, s71\n\t"
            "s_cmp_eq_u64 s[86:87], 0\n\t"
            "s_cbranch_scc1 PKA_NONEAR_%=\n\t"
            "s_cmp_eq_u64 s[88:89], 0\n\t"
            "s_cbranch_scc1 PKA_NOFAR_%=\n\t"
            "s_cmp_ge_i32 %[sp], 64\n\t"
            "s_cbranch_scc1 PKA_OVER_%=\n\t"
            "s_mov_b32 m0, %[sp]\n\t"
            "s_nop 0\n\t"
            "v_writelane_b32 %[sta], s92, m0\n\t"
            "v_writelane_b32 %[stb], s93, m0\n\t"
            "v_writelane_b32 %[stmlo], s88, m0\n\t"
            "v_writelane_b32 %[stmhi], s89, m0\n\t"
            "v_writelane_b32 %[stdf], %[depth], m0\n\t"
            "s_add_i32 %[sp], %[sp], 1\n\t"
            "PKA_NOFAR_%=:\n\t"
            "s_mov_b32 %[first], s80\n\t"
            "s_mov_b32 %[cnt], s81\n\t"
            "s_mov_b64 %[m], s[86:87]\n\t"
            "s_add_i32 %[depth], %[depth], 1\n\t"
            "s_branch PKA_CHECK_%=\n\t"
            "PKA_OVER_%=:\n\t"
            "s_mov_b32 %[err], 1\n\t"
            "s_branch PKA_NOFAR_%=\n\t"
            "PKA_NONEAR_%=:\n\t"
            "s_mov_b32 %[first], s92\n\t"
            "s_mov_b32 %[cnt], s93\n\t"
            "s_mov_b64 %[m], s[88:89]\n\t"
            "s_cmp_eq_u64 s[88:89], 0\n\t"
            "s_cbranch_scc1 PKA_END_%=\n\t"
            "PKA_CHECK_%=:\n\t"
            "s_and_b32 s90, %[cnt], 0x3fffffff\n\t"
            "s_cmp_eq_u32 s90, 0\n\t"
            "s_cbranch_scc1 PKA_TOP_%=\n\t"
            "PKA_END_%=:\n\t"
            : [first] "+&s"(W.cur_first), [cnt] "+&s"(W.cur_cnt), [m] "+&s"(W.m), [sp] "+&s"(W.sp), [depth] "+&s"(W.depth), [err] "+&s"(W.err),
              [sta] "+&v"(W.st_a), [stb] "+&v"(W.st_b), [stmlo] "+&v"(W.st_mlo), [stmhi] "+&v"(W.st_mhi), [stdf] "+&v"(W.st_df)
            : [nodes] "s"(nodes), [oxy] "v"(W.ray.o_xy), [ozz] "v"(W.ray.o_zz), [ixy] "v"(W.ray.i_xy), [izz] "v"(W.ray.i_zz), [tcur] "v"(W.ray.tcur),
              [posx] "s"(posx), [posy] "s"(posy), [posz] "s"(posz), [ssz] "s"(stack_size), [eps] "s"(eps), [ordered] "s"(ordered)
            : "vcc", "scc", "m0",
              "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
              "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",
              "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    } else {
        asm volatile(
            "s_mov_b64 s[94:95], %[nodes]\n\t"
            "PKC_TOP_%=:\n\t"
            "s_add_i32 s90, %[depth], 2\n\t"
            "s_cmp_gt_i32 s90, %[ssz]\n\t"
            "s_cselect_b32 %[err], 1, %[err]\n\t"
            "s_mov_b32 s92, %[first]\n\t"
            "s_mov_b32 s93, 0\n\t"
            "s_lshl_b64 s[92:93], s[92:93], 5\n\t"
            "s_add_u32 s92, s92, s94\n\t"
            "s_addc_u32 s93, s93, s95\n\t"
            "s_load_dwordx16 s[64:79], s[92:93], 0x0\n\t"
            // order (closest hit): every lane must see ITS OWN near / far order; lanes that disagree are parked on this node
            "s_mov_b32 s91, 1\n\t"
            "s_cmp_eq_u32 %[ordered], 0\n\t"
            "s_cbranch_scc1 PKC_ORD_%=\n\t"
            "s_lshr_b32 s90, %[cnt], 30\n\t"
            "s_cmp_eq_u32 s90, 0\n\t"
            "s_cbranch_scc1 PKC_ORD_%=\n\t"
            "s_mov_b64 s[86:87], %[posx]\n\t"
            "s_cmp_eq_u32 s90, 2\n\t"
            "s_cselect_b64 s[86:87], %[posy], s[86:87]\n\t"
            "s_cmp_eq_u32 s90, 3\n\t"
            "s_cselect_b64 s[86:87], %[posz], s[86:87]\n\t"
            "s_and_b64 s[86:87], s[86:87], %[m]\n\t"
            "s_cmp_eq_u64 s[86:87], 0\n\t"
            "s_cselect_b32 s91, 0, 1\n\t"
            "s_cbranch_scc1 PKC_ORD_%=\n\t"
            "s_cmp_eq_u64 s[86:87], %[m]\n\t"
            "s_cbranch_scc1 PKC_ORD_%=\n\t"
            // split: park (node, m & ~pos) as a pretested entry, continue with the left-first lanes
            "s_andn2_b64 s[88:89], %[m], s[86:87]\n\t"
            "s_mov_b64 %[m], s[86:87]\n\t"
            "s_cmp_ge_i32 %[sp], 64\n\t"
            "s_cselect_b32 %[err], 1, %[err]\n\t"
            "s_cbranch_scc1 PKC_ORD_%=\n\t"
            "s_mov_b32 m0, %[sp]\n\t"
            "s_or_b32 s90, %[depth], 0x20000000\n\t"
            "v_writelane_b32 %[sta], %[first], m0\n\t"
            "v_writelane_b32 %[stb], %[cnt], m0\n\t"
            "v_writelane_b32 %[stmlo], s88, m0\n\t"
            "v_writelane_b32 %[stmhi], s89, m0\n\t"
            "v_writelane_b32 %[stdf], s90, m0\n\t"
            "v_writelane_b32 %[stidx], %[idx], m0\n\t"
            "s_add_i32 %[sp], %[sp], 1\n\t"
            "PKC_ORD_%=:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[82:83]")
            PK_ASM_SLAB(72, 73, 74, 75, 76, 77, "s[84:85]")
            // near / far (left child index = first, right = first + 1)
            "s_add_i32 s90, %[first], 1\n\t"
            "s_cmp_lg_u32 s91, 0\n\t"
            "s_cselect_b64 s[86:87], s[82:83], s[84:85]\n\t"
            "s_cselect_b64 s[88:89], s[84:85], s[82:83]\n\t"
            "s_cselect_b32 s80, s70, s78\n\t"
            "s_cselect_b32 s81, s71, s79\n\t"
            "s_cselect_b32 s92, s78, s70\n\t"
            "s_cselect_b32 s93, s79, s71\n\t"
            "s_cselect_b32 s82, %[first], s90\n\t"          // near index
            "s_cselect_b32 s83, s90, %[first]\n\t"          // far index
            "s_cmp_eq_u64 s[86:87], 0\n\t"
            "s_cbranch_scc1 PKC_NONEAR_%=\n\t"
            "s_cmp_eq_u64 s[88:89], 0\n\t"
            "s_cbranch_scc1 PKC_NOFAR_%=\n\t"
            "s_cmp_ge_i32 %[sp], 64\n\t"
            "s_cbranch_scc1 PKC_OVER_%=\n\t"
            "s_mov_b32 m0, %[sp]\n\t"
            "s_nop 0\n\t"
            "v_writelane_b32 %[sta], s92, m0\n\t"
            "v_writelane_b32 %[stb], s93, m0\n\t"
            "v_writelane_b32 %[stmlo], s88, m0\n\t"
            "v_writelane_b32 %[stmhi], s89, m0\n\t"
            "v_writelane_b32 %[stdf], %[depth], m0\n\t"
            "v_writelane_b32 %[stidx], s83, m0\n\t"
            "s_add_i32 %[sp], %[sp], 1\n\t"
            "PKC_NOFAR_%=:\n\t"
            "s_mov_b32 %[first], s80\n\t"
            "s_mov_b32 %[cnt], s81\n\t"
            "s_mov_b32 %[idx], s82\n\t"
            "s_mov_b64 %[m], s[86:87]\n\t"
            "s_add_i32 %[depth], %[depth], 1\n\t"
            "s_branch PKC_CHECK_%=\n\t"
            "PKC_OVER_%=:\n\t"
            "s_mov_b32 %[err], 1\n\t"
            "s_branch PKC_NOFAR_%=\n\t"
            "PKC_NONEAR_%=:\n\t"
            "s_mov_b32 %[first], s92\n\t"
            "s_mov_b32 %[cnt], s93\n\t"
            "s_mov_b32 %[idx], s83\n\t"
            "s_mov_b64 %[m], s[88:89]\n\t"
            "s_cmp_eq_u64 s[88:89], 0\n\t"
            "s_cbranch_scc1 PKC_END_%=\n\t"
            "PKC_CHECK_%=:\n\t"
            "s_add_i32 %[steps], %[steps], 1\n\t"
            "s_bcnt1_i32_b64 s90, %[m]\n\t"               // hybrid: a node wanted by few lanes goes back to the caller (lanes turn private)
            "s_cmp_le_i32 s90, %[thr]\n\t"
            "s_cbranch_scc1 PKC_END_%=\n\t"
            "s_and_b32 s90, %[cnt], 0x3fffffff\n\t"
            "s_cmp_eq_u32 s90, 0\n\t"
            "s_cbranch_scc1 PKC_TOP_%=\n\t"
            "PKC_END_%=:\n\t"
            : [first] "+&s"(W.cur_first), [cnt] "+&s"(W.cur_cnt), [idx] "+&s"(W.cur_idx), [m] "+&s"(W.m), [sp] "+&s"(W.sp), [depth] "+&s"(W.depth), [err] "+&s"(W.err),
              [steps] "+&s"(steps),
              [sta] "+&v"(W.st_a), [stb] "+&v"(W.st_b), [stmlo] "+&v"(W.st_mlo), [stmhi] "+&v"(W.st_mhi), [stdf] "+&v"(W.st_df), [stidx] "+&v"(W.st_idx)
            : [nodes] "s"(nodes), [oxy] "v"(W.ray.o_xy), [ozz] "v"(W.ray.o_zz), [ixy] "v"(W.ray.i_xy), [izz] "v"(W.ray.i_zz), [tcur] "v"(W.ray.tcur),
              [posx] "s"(posx), [posy] "s"(posy), [posz] "s"(posz), [ssz] "s"(stack_size), [eps] "s"(eps), [ordered] "s"(ordered), [thr] "s"(thr)
            : "vcc", "scc", "m0",
              "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
              "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",
              "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    }
    W.pk_steps = steps;
}

// triangles s98 .. s99-1 of W.tris against the lanes of m: triangle_intersect (BottomLevelBVH.cpp:311-346), the triangle in SGPRs
// (labels PKB_TRI / PKB_NEXT / PKB_POP and the operands belong to the enclosing asm statement; s88 / s89 / s[84:87] are temporaries)
#define PK_ASM_TRI_LOOP \
        "PKB_TRI_%=:\n\t"                                                                                                         \
        "s_lshl_b32 s89, s98, 6\n\t"                                                                                              \
        "s_load_dwordx8 s[64:71], s[96:97], s89\n\t"                                                                              \
        "s_add_u32 s89, s89, 32\n\t"                                                                                              \
        "s_load_dwordx4 s[72:75], s[96:97], s89\n\t"                                                                              \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                \
        /* h = cross(d, e2)                                      p0 = s64..66, e1 = s68..70, e2 = s72..74 */                      \
        "v_mul_f32 %[t0], s74, %[dy]\n\t"                                                                                         \
        "v_mul_f32 %[t1], s73, %[dz]\n\t"                                                                                         \
        "v_mul_f32 %[t2], s72, %[dz]\n\t"                                                                                         \
        "v_mul_f32 %[t3], s74, %[dx]\n\t"                                                                                         \
        "v_mul_f32 %[t4], s73, %[dx]\n\t"                                                                                         \
        "v_mul_f32 %[t5], s72, %[dy]\n\t"                                                                                         \
        "v_sub_f32 v24, %[t0], %[t1]\n\t"                    /* h.x */                                                            \
        "v_sub_f32 v25, %[t2], %[t3]\n\t"                    /* h.y */                                                            \
        "v_sub_f32 v26, %[t4], %[t5]\n\t"                    /* h.z */                                                            \
        /* a = e1.x*h.x + (e1.y*h.y + e1.z*h.z) */                                                                                \
        "v_mul_f32 %[t0], s69, v25\n\t"                                                                                           \
        "v_mul_f32 %[t1], s70, v26\n\t"                                                                                           \
        "v_mul_f32 %[t2], s68, v24\n\t"                                                                                           \
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"                                                                                       \
        "v_add_f32 %[t9], %[t2], %[t0]\n\t"                  /* a */                                                              \
        /* f = 1.0f / a, correctly rounded (the expansion hipcc emits for IEEE division with denormals enabled) */                \
        "v_div_scale_f32 %[t0], s[84:85], %[t9], %[t9], 1.0\n\t"                                                                  \
        "v_rcp_f32 %[t1], %[t0]\n\t"                                                                                              \
        "v_div_scale_f32 %[t2], vcc, 1.0, %[t9], 1.0\n\t"                                                                         \
        /* s = o - p0 (independent work in the shadow of the rcp) */                                                              \
        "v_subrev_f32 v27, s64, %[ox]\n\t"                                                                                        \
        "v_fma_f32 %[t3], -%[t0], %[t1], 1.0\n\t"                                                                                 \
        "v_fmac_f32 %[t1], %[t3], %[t1]\n\t"                                                                                      \
        "v_mul_f32 %[t4], %[t2], %[t1]\n\t"                                                                                       \
        "v_fma_f32 %[t5], -%[t0], %[t4], %[t2]\n\t"                                                                               \
        "v_fmac_f32 %[t4], %[t5], %[t1]\n\t"                                                                                      \
        "v_fma_f32 %[t0], -%[t0], %[t4], %[t2]\n\t"                                                                               \
        "v_subrev_f32 v28, s65, %[oy]\n\t"                                                                                        \
        "v_div_fmas_f32 %[t0], %[t0], %[t1], %[t4]\n\t"                                                                           \
        "v_subrev_f32 v29, s66, %[oz]\n\t"                                                                                        \
        "v_div_fixup_f32 %[t8], %[t0], %[t9], 1.0\n\t"       /* f */                                                              \
        /* u = f * (s.x*h.x + (s.y*h.y + s.z*h.z)) */                                                                             \
        "v_mul_f32 %[t0], v28, v25\n\t"                                                                                           \
        "v_mul_f32 %[t1], v29, v26\n\t"                                                                                           \
        "v_mul_f32 %[t2], v27, v24\n\t"                                                                                           \
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"                                                                                       \
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"                                                                                       \
        "v_mul_f32 %[t7], %[t8], %[t0]\n\t"                  /* u */                                                              \
        "v_cmp_lt_f32 s[84:85], 0, %[t7]\n\t"                /* u > 0 */                                                          \
        "v_cmp_gt_f32 s[86:87], 1.0, %[t7]\n\t"              /* u < 1 */                                                          \
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"                                                                              \
        "s_and_b64 s[84:85], s[84:85], %[m]\n\t"                                                                                  \
        "s_cmp_lg_u64 s[84:85], 0\n\t"                                                                                            \
        "s_cbranch_scc0 PKB_NEXT_%=\n\t"                     /* no lane of the node passes the u test */                          \
        /* q = cross(s, e1) */                                                                                                    \
        "v_mul_f32 %[t0], s70, v28\n\t"                                                                                           \
        "v_mul_f32 %[t1], s69, v29\n\t"                                                                                           \
        "v_mul_f32 %[t2], s68, v29\n\t"                                                                                           \
        "v_mul_f32 %[t3], s70, v27\n\t"                                                                                           \
        "v_mul_f32 %[t4], s69, v27\n\t"                                                                                           \
        "v_mul_f32 %[t5], s68, v28\n\t"                                                                                           \
        "v_sub_f32 v24, %[t0], %[t1]\n\t"                    /* q.x  (h is dead) */                                               \
        "v_sub_f32 v25, %[t2], %[t3]\n\t"                    /* q.y */                                                            \
        "v_sub_f32 v26, %[t4], %[t5]\n\t"                    /* q.z */                                                            \
        /* v = f * (d.x*q.x + (d.y*q.y + d.z*q.z)) */                                                                             \
        "v_mul_f32 %[t0], %[dy], v25\n\t"                                                                                         \
        "v_mul_f32 %[t1], %[dz], v26\n\t"                                                                                         \
        "v_mul_f32 %[t2], %[dx], v24\n\t"                                                                                         \
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"                                                                                       \
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"                                                                                       \
        "v_mul_f32 %[t6], %[t8], %[t0]\n\t"                  /* v */                                                              \
        /* t = f * (e2.x*q.x + (e2.y*q.y + e2.z*q.z)) */                                                                          \
        "v_mul_f32 %[t0], s73, v25\n\t"                                                                                           \
        "v_mul_f32 %[t1], s74, v26\n\t"                                                                                           \
        "v_mul_f32 %[t2], s72, v24\n\t"                                                                                           \
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"                                                                                       \
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"                                                                                       \
        "v_mul_f32 %[t5], %[t8], %[t0]\n\t"                  /* t */                                                              \
        "v_add_f32 %[t4], %[t7], %[t6]\n\t"                  /* u + v */                                                          \
        "v_cmp_lt_f32 s[86:87], 0, %[t6]\n\t"                /* v > 0 */                                                          \
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"                                                                              \
        "v_cmp_gt_f32 s[86:87], 1.0, %[t4]\n\t"              /* u + v < 1 */                                                      \
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"                                                                              \
        "v_cmp_lt_f32 s[86:87], %[eps], %[t5]\n\t"           /* t > EPSILON */                                                    \
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"                                                                              \
        "v_cmp_lt_f32 s[86:87], %[t5], %[tcur]\n\t"          /* t < max distance */                                               \
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"                                                                              \
        /* lanes that are hit leave the packet */                                                                                 \
        "s_andn2_b64 %[alive], %[alive], s[84:85]\n\t"                                                                            \
        "s_andn2_b64 %[m], %[m], s[84:85]\n\t"                                                                                    \
        "s_cmp_lg_u64 %[m], 0\n\t"                                                                                                \
        "s_cbranch_scc0 PKB_POP_%=\n\t"                                                                                           \
        "PKB_NEXT_%=:\n\t"                                                                                                        \
        "s_add_i32 s98, s98, 1\n\t"                                                                                               \
        "s_cmp_lt_i32 s98, s99\n\t"                                                                                               \
        "s_cbranch_scc1 PKB_TRI_%=\n\t"

// ---- the whole any-hit walk of one BLAS, hand-scheduled ------------------------------------------------------------------------
// BottomLevelBVH::intersect (BottomLevelBVH.cpp:398-437) for the lanes of W.m, from the node in W.cur_* until the packet stack is back
// at W.floor_sp or no lane is alive: inner nodes (child-pair fetch, two slab tests, push of the right child), leaves
// (triangle_intersect, BottomLevelBVH.cpp:311-346, with the triangle in SGPRs) and pops in ONE asm block.  Any-hit results do not
// depend on the visiting order, so the left child is always walked first (no order logic at all).  Occlusion is pure mask
// arithmetic: a lane that is hit leaves W.alive (the caller derives the per-lane flag from the masks).
//   stack entry: a = first, b = count | depth << 16 (leaf counts < 65536 and node / triangle offsets < 2^31 are checked at upload)
//   SGPRs: s[64:79] child pair / triangle, s[80:83] child masks, s[84:93] temporaries, s[94:97] node and triangle base addresses,
//   s98/s99 triangle loop; VGPRs: v[24:31] slab pairs / temporaries, the rest are compiler-allocated operands.
//   Instruction budget per inner step: 34 VALU (+4 v_writelane on a push), 16-24 SALU.
//   Hazards (gfx940/950): v_rcp_f32 (trans) result is used two instructions later; v_div_scale writes VCC four instructions before
//   v_div_fmas reads it; SGPRs written by v_cmp are read by SALU only; v_readlane results are consumed by SALU / SMEM (interlocked).
RTX_D void pk_blas_any_asm(PkWalk & W, int stack_size, const uint32_t lane_lds, int defer_t0, int grow_shift) {
    W.cur_first = rfl(W.cur_first); W.cur_cnt = rfl(W.cur_cnt); W.m = rfl64(W.m); W.alive = rfl64(W.alive); W.sp = rfl(W.sp);
    int depth = rfl(W.depth), maxd = -1, err = 0, steps = rfl(W.pk_steps);
    const int floor_sp = rfl(W.floor_sp);
    defer_t0 = rfl(defer_t0); grow_shift = rfl(grow_shift);
    const float eps = RAY_EPSILON;
    const unsigned long long nodes = rfl64((unsigned long long)W.nodes), tris = rfl64((unsigned long long)W.tris);
    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;          // triangle-test temporaries
    asm volatile(
        "s_mov_b64 s[94:95], %[nodes]\n\t"
        "s_mov_b64 s[96:97], %[tris]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKB_POP_%=\n\t"
        "PKB_DISPATCH_%=:\n\t"
        // hybrid: a node wanted by few lanes is handed to those lanes' private stacks (walked per lane after the packet walk);
        // the threshold grows with the length of the walk, so a packet that turns out incoherent converts itself to per-lane work
        "s_add_i32 %[steps], %[steps], 1\n\t"
        "s_bcnt1_i32_b64 s89, %[m]\n\t"
        "s_lshr_b32 s90, %[steps], %[gsh]\n\t"
        "s_add_i32 s90, s90, %[t0d]\n\t"
        "s_cmp_le_u32 s89, s90\n\t"
        "s_cbranch_scc1 PKB_DEFER_%=\n\t"
        "PKB_NODEFER_%=:\n\t"
        "s_and_b32 s88, %[cnt], 0xffff\n\t"
        "s_cmp_lg_u32 s88, 0\n\t"
        "s_cbranch_scc1 PKB_LEAF_%=\n\t"
        // ---------------- inner node ----------------
        "PKB_INNER_%=:\n\t"
        "s_lshl_b32 s88, %[first], 5\n\t"
        "s_load_dwordx16 s[64:79], s[94:95], s88\n\t"
        "s_max_i32 %[maxd], %[maxd], %[depth]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[80:81]")
        PK_ASM_SLAB(72, 73, 74, 75, 76, 77, "s[82:83]")
        "s_cmp_lg_u64 s[80:81], 0\n\t"
        "s_cbranch_scc0 PKB_NOLEFT_%=\n\t"
        "s_cmp_lg_u64 s[82:83], 0\n\t"
        "s_cbranch_scc0 PKB_GOLEFT_%=\n\t"
        // push the right child (popped with `depth` entries pending)
        "s_cmp_ge_i32 %[sp], 64\n\t"
        "s_cbranch_scc1 PKB_OVER_%=\n\t"
        "s_lshl_b32 s88, %[depth], 16\n\t"
        "s_or_b32 s88, s88, s79\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "s_add_i32 %[sp], %[sp], 1\n\t"
        "v_writelane_b32 %[sta], s78, m0\n\t"
        "v_writelane_b32 %[stb], s88, m0\n\t"
        "v_writelane_b32 %[stmlo], s82, m0\n\t"
        "v_writelane_b32 %[stmhi], s83, m0\n\t"
        "PKB_GOLEFT_%=:\n\t"
        "s_mov_b32 %[first], s70\n\t"
        "s_mov_b32 %[cnt], s71\n\t"
        "s_mov_b64 %[m], s[80:81]\n\t"
        "s_add_i32 %[depth], %[depth], 1\n\t"
        "s_branch PKB_DISPATCH_%=\n\t"
        "PKB_OVER_%=:\n\t"
        "s_mov_b32 %[err], 1\n\t"
        "s_branch PKB_GOLEFT_%=\n\t"
        "PKB_NOLEFT_%=:\n\t"
        "s_cmp_lg_u64 s[82:83], 0\n\t"
        "s_cbranch_scc0 PKB_POP_%=\n\t"
        "s_mov_b32 %[first], s78\n\t"
        "s_mov_b32 %[cnt], s79\n\t"
        "s_mov_b64 %[m], s[82:83]\n\t"
        "s_branch PKB_DISPATCH_%=\n\t"
        // ---------------- defer: push the node on the private stacks of the lanes in m ----------------
        "PKB_DEFER_%=:\n\t"
        "s_and_b32 s88, %[cnt], 0xffff\n\t"
        "s_cmp_ge_u32 s88, 16\n\t"                          // packable entry: leaf count < 16, first < 2^24
        "s_cbranch_scc1 PKB_NODEFER_%=\n\t"
        "s_cmp_ge_u32 %[first], 0x1000000\n\t"
        "s_cbranch_scc1 PKB_NODEFER_%=\n\t"
        "v_cmp_le_u32 vcc, 8, %[psp]\n\t"                   // every lane of m needs room on its private stack
        "s_and_b64 s[84:85], vcc, %[m]\n\t"
        "s_cmp_lg_u64 s[84:85], 0\n\t"
        "s_cbranch_scc1 PKB_NODEFER_%=\n\t"
        "s_lshr_b32 s89, %[cnt], 30\n\t"
        "s_lshl_b32 s89, s89, 28\n\t"
        "s_lshl_b32 s90, s88, 24\n\t"
        "s_or_b32 s89, s89, s90\n\t"
        "s_or_b32 s89, s89, %[first]\n\t"
        "s_or_b32 s89, s89, 0x40000000\n\t"
        "s_mov_b64 s[84:85], exec\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_lshl_add_u32 %[t0], %[psp], 8, %[llds]\n\t"
        "v_mov_b32 %[t1], s89\n\t"
        "ds_write_b32 %[t0], %[t1]\n\t"
        "v_add_u32 %[psp], 1, %[psp]\n\t"
        "s_mov_b64 exec, s[84:85]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_branch PKB_POP_%=\n\t"
        // ---------------- leaf: triangles first .. first + count - 1 ----------------
        "PKB_LEAF_%=:\n\t"
        "s_mov_b32 s98, %[first]\n\t"
        "s_add_i32 s99, %[first], s88\n\t"
        PK_ASM_TRI_LOOP
        // ---------------- pop ----------------
        "PKB_POP_%=:\n\t"
        "s_mov_b64 %[m], 0\n\t"
        "s_cmp_eq_u64 %[alive], 0\n\t"
        "s_cbranch_scc1 PKB_END_%=\n\t"
        "s_cmp_le_i32 %[sp], %[floor]\n\t"
        "s_cbranch_scc1 PKB_END_%=\n\t"
        "s_sub_i32 %[sp], %[sp], 1\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "v_readlane_b32 s84, %[stmlo], m0\n\t"
        "v_readlane_b32 s85, %[stmhi], m0\n\t"
        "v_readlane_b32 %[first], %[sta], m0\n\t"
        "v_readlane_b32 s88, %[stb], m0\n\t"
        "s_and_b64 %[m], s[84:85], %[alive]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKB_POP_%=\n\t"
        "s_bfe_u32 %[depth], s88, 0xe0010\n\t"               // depth = (b >> 16) & 0x3fff
        "s_and_b32 %[cnt], s88, 0xc000ffff\n\t"
        "s_branch PKB_DISPATCH_%=\n\t"
        "PKB_END_%=:\n\t"
        : [first] "+&s"(W.cur_first), [cnt] "+&s"(W.cur_cnt), [m] "+&s"(W.m), [alive] "+&s"(W.alive), [sp] "+&s"(W.sp), [depth] "+&s"(depth),
          [maxd] "+&s"(maxd), [err] "+&s"(err), [steps] "+&s"(steps),
          [sta] "+&v"(W.st_a), [stb] "+&v"(W.st_b), [stmlo] "+&v"(W.st_mlo), [stmhi] "+&v"(W.st_mhi), [psp] "+&v"(W.psp),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7),
          [t8] "=&v"(t8), [t9] "=&v"(t9)
        : [nodes] "s"(nodes), [tris] "s"(tris), [floor] "s"(floor_sp),
          [oxy] "v"(W.ray.o_xy), [ozz] "v"(W.ray.o_zz), [ixy] "v"(W.ray.i_xy), [izz] "v"(W.ray.i_zz), [tcur] "v"(W.ray.tcur),
          [ox] "v"(W.ray.o_xy.x), [oy] "v"(W.ray.o_xy.y), [oz] "v"(W.ray.o_zz.x), [dx] "v"(W.ray.d.x), [dy] "v"(W.ray.d.y), [dz] "v"(W.ray.d.z),
          [eps] "s"(eps), [t0d] "s"(defer_t0), [gsh] "s"(grow_shift), [llds] "v"(lane_lds)
        : "vcc", "scc", "m0", "memory",
          "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    W.depth = depth; W.pk_steps = steps;
    if (maxd + 2 > stack_size || err) W.err = 1;       // reference stack: `depth` pending + 2 pushed at the deepest inner node expanded
}

// ---- the same walk over the 4-wide records (rtx_api.hip, build_nodes_pk4) ----------------------------------------------------------
// For meshes whose boxes are nested (every reference-built tree): the record of an inner node lists its grandchildren, so one fetch
// (2 x s_load_dwordx16) feeds four slab tests and the walk takes about half the dependent steps of the binary one; why skipping the
// intermediate boxes cannot change an any-hit answer is argued at build_nodes_pk4.  Slot 0 is walked first, the other passing slots are
// pushed (3, 2, 1: popped in slot order); the host has bounded the stack need of every path, so there is no depth bookkeeping.
//   stack entry: a = first, b = leaf count; deferred entry: leaf count << 24 | first (counts < 256, offsets < 2^24: checked at upload)
//   SGPRs: s[48:79] the record (slot k at s[48+8k : 55+8k]) / the triangle, s[80:87] slot masks, s88-s93 temporaries, s[94:97] bases.
//   Instruction budget per step: 68 VALU (+4 v_writelane per push), 20-45 SALU.
#define RTX_PK4_MAX_NEED 36          // packet-stack entries a BLAS walk may need (64 lanes hold the stack; the TLAS part keeps the rest)
#define PK4_ASM_PUSH(K)                                                                                                             \
        "s_cmp_lg_u64 %[m], 0\n\t"                                                                                                  \
        "s_cbranch_scc0 PKB_SET" #K "_%=\n\t"                                                                                       \
        "s_cmp_ge_i32 %[sp], 64\n\t"                                                                                                \
        "s_cbranch_scc1 PKB_OVER" #K "_%=\n\t"                                                                                      \
        "s_mov_b32 m0, %[sp]\n\t"                                                                                                   \
        "s_mov_b64 s[92:93], %[m]\n\t"                                                                                              \
        "s_add_i32 %[sp], %[sp], 1\n\t"                                                                                             \
        "v_writelane_b32 %[sta], %[first], m0\n\t"                                                                                  \
        "v_writelane_b32 %[stb], %[cnt], m0\n\t"                                                                                    \
        "v_writelane_b32 %[stmlo], s92, m0\n\t"                                                                                     \
        "v_writelane_b32 %[stmhi], s93, m0\n\t"                                                                                     \
        "PKB_SET" #K "_%=:\n\t"
RTX_D void pk_blas_any_asm4(PkWalk & W, rtx_cptr nodes4, const uint32_t lane_lds, int defer_t0, int defer_leaf, int grow_shift,
                              uint32_t item_cap, unsigned long long item_base, uint32_t slot) {
    W.cur_first = rfl(W.cur_first); W.cur_cnt = rfl(W.cur_cnt & 0x3fffffff); W.m = rfl64(W.m); W.alive = rfl64(W.alive); W.sp = rfl(W.sp);
    int err = 0, steps = rfl(W.pk_steps);
    const int floor_sp = rfl(W.floor_sp);
    defer_t0 = rfl(defer_t0); defer_leaf = rfl(defer_leaf); grow_shift = rfl(grow_shift);
    // do the lanes of the walk agree on the signs of the inverse direction?  (lanes outside W.m never join later: masks only shrink)
    int sgn = 0;
    {
        const unsigned long long nx = __ballot(W.ray.i_xy.x < 0.0f) & W.m, ny = __ballot(W.ray.i_xy.y < 0.0f) & W.m, nz = __ballot(W.ray.i_zz.x < 0.0f) & W.m;
        if ((nx != 0ull && nx != W.m) || (ny != 0ull && ny != W.m) || (nz != 0ull && nz != W.m)) sgn = 8;
        else sgn = (nx != 0ull ? 1 : 0) | (ny != 0ull ? 2 : 0) | (nz != 0ull ? 4 : 0);
        sgn = rfl(sgn);
    }
    int icnt = rfl(W.item_n); const int icap = rfl((int)item_cap), inst = rfl(W.cur_blas); item_base = rfl64(item_base);
    const float eps = RAY_EPSILON;
    const unsigned long long nodes = rfl64((unsigned long long)nodes4), tris = rfl64((unsigned long long)W.tris);
    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;          // triangle-test temporaries
    asm volatile(
        "s_mov_b64 s[94:95], %[nodes]\n\t"
        "s_mov_b64 s[96:97], %[tris]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKB_POP_%=\n\t"
        "PKB_DISPATCH_%=:\n\t"
        "s_add_i32 %[steps], %[steps], 1\n\t"
        "s_bcnt1_i32_b64 s89, %[m]\n\t"
        "s_lshr_b32 s90, %[steps], %[gsh]\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"                       // a leaf costs the packet a triangle test per triangle: its own threshold
        "s_cselect_b32 s91, %[tld], %[t0d]\n\t"
        "s_add_i32 s90, s90, s91\n\t"
        "s_cmp_le_u32 s89, s90\n\t"
        "s_cbranch_scc1 PKB_DEFER_%=\n\t"
        "PKB_NODEFER_%=:\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 PKB_LEAF_%=\n\t"
        // ---------------- inner node: four slots ----------------
        "s_lshl_b32 s88, %[first], 6\n\t"
        "s_load_dwordx16 s[48:63], s[94:95], s88\n\t"
        "s_add_u32 s88, s88, 64\n\t"
        "s_load_dwordx16 s[64:79], s[94:95], s88\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // sign-coherent packets (sgn = bit 0 / 1 / 2: inv.x / .y / .z negative for every lane; 8: lanes disagree) take the 11-instruction test
        "s_cmp_ge_u32 %[sgn], 8\n\t"
        "s_cbranch_scc1 PKB_VG_%=\n\t"
        "s_bitcmp1_b32 %[sgn], 2\n\t"
        "s_cbranch_scc1 PKB_T4_%=\n\t"
        "s_bitcmp1_b32 %[sgn], 1\n\t"
        "s_cbranch_scc1 PKB_T2_%=\n\t"
        "s_bitcmp1_b32 %[sgn], 0\n\t"
        "s_cbranch_scc1 PKB_V1_%=\n\t"
        PK_ASM_QUAD_S("v24", "v26", "v25", "v27", "v28", "v29")      // + + +
        "PKB_V1_%=:\n\t"
        PK_ASM_QUAD_S("v26", "v24", "v25", "v27", "v28", "v29")      // - + +
        "PKB_T2_%=:\n\t"
        "s_bitcmp1_b32 %[sgn], 0\n\t"
        "s_cbranch_scc1 PKB_V3_%=\n\t"
        PK_ASM_QUAD_S("v24", "v26", "v27", "v25", "v28", "v29")      // + - +
        "PKB_V3_%=:\n\t"
        PK_ASM_QUAD_S("v26", "v24", "v27", "v25", "v28", "v29")      // - - +
        "PKB_T4_%=:\n\t"
        "s_bitcmp1_b32 %[sgn], 1\n\t"
        "s_cbranch_scc1 PKB_T6_%=\n\t"
        "s_bitcmp1_b32 %[sgn], 0\n\t"
        "s_cbranch_scc1 PKB_V5_%=\n\t"
        PK_ASM_QUAD_S("v24", "v26", "v25", "v27", "v29", "v28")      // + + -
        "PKB_V5_%=:\n\t"
        PK_ASM_QUAD_S("v26", "v24", "v25", "v27", "v29", "v28")      // - + -
        "PKB_T6_%=:\n\t"
        "s_bitcmp1_b32 %[sgn], 0\n\t"
        "s_cbranch_scc1 PKB_V7_%=\n\t"
        PK_ASM_QUAD_S("v24", "v26", "v27", "v25", "v29", "v28")      // + - -
        "PKB_V7_%=:\n\t"
        PK_ASM_QUAD_S("v26", "v24", "v27", "v25", "v29", "v28")      // - - -
        "PKB_VG_%=:\n\t"
        PK_ASM_SLAB(48, 49, 50, 51, 52, 53, "s[80:81]")
        PK_ASM_SLAB(56, 57, 58, 59, 60, 61, "s[82:83]")
        PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[84:85]")
        PK_ASM_SLAB(72, 73, 74, 75, 76, 77, "s[86:87]")
        "PKB_SEL_%=:\n\t"
        // the lowest passing slot is walked next, the others are pushed
        "s_mov_b64 %[m], 0\n\t"
        "s_cmp_lg_u64 s[86:87], 0\n\t"
        "s_cbranch_scc0 PKB_S2_%=\n\t"
        "s_mov_b32 %[first], s78\n\t"
        "s_mov_b32 %[cnt], s79\n\t"
        "s_mov_b64 %[m], s[86:87]\n\t"
        "PKB_S2_%=:\n\t"
        "s_cmp_lg_u64 s[84:85], 0\n\t"
        "s_cbranch_scc0 PKB_S1_%=\n\t"
        PK4_ASM_PUSH(2)
        "s_mov_b32 %[first], s70\n\t"
        "s_mov_b32 %[cnt], s71\n\t"
        "s_mov_b64 %[m], s[84:85]\n\t"
        "PKB_S1_%=:\n\t"
        "s_cmp_lg_u64 s[82:83], 0\n\t"
        "s_cbranch_scc0 PKB_S0_%=\n\t"
        PK4_ASM_PUSH(1)
        "s_mov_b32 %[first], s62\n\t"
        "s_mov_b32 %[cnt], s63\n\t"
        "s_mov_b64 %[m], s[82:83]\n\t"
        "PKB_S0_%=:\n\t"
        "s_cmp_lg_u64 s[80:81], 0\n\t"
        "s_cbranch_scc0 PKB_SDONE_%=\n\t"
        PK4_ASM_PUSH(0)
        "s_mov_b32 %[first], s54\n\t"
        "s_mov_b32 %[cnt], s55\n\t"
        "s_mov_b64 %[m], s[80:81]\n\t"
        "PKB_SDONE_%=:\n\t"
        "s_cmp_lg_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKB_DISPATCH_%=\n\t"
        "s_branch PKB_POP_%=\n\t"
        "PKB_OVER2_%=:\n\t"
        "s_mov_b32 %[err], 1\n\t"
        "s_branch PKB_SET2_%=\n\t"
        "PKB_OVER1_%=:\n\t"
        "s_mov_b32 %[err], 1\n\t"
        "s_branch PKB_SET1_%=\n\t"
        "PKB_OVER0_%=:\n\t"
        "s_mov_b32 %[err], 1\n\t"
        "s_branch PKB_SET0_%=\n\t"
        // ---------------- defer: push the node on the private stacks of the lanes in m ----------------
        "PKB_DEFER_%=:\n\t"
        "s_cmp_eq_u32 %[icap], 0\n\t"
        "s_cbranch_scc1 PKB_DEFER_LDS_%=\n\t"
        // split walk: items (model-space ray, ray slot, node, mesh; 48 bytes) appended to this wave's own chunk — no atomics; k_items walks them one per lane
        "s_add_u32 s88, %[icnt], 64\n\t"
        "s_cmp_gt_u32 s88, %[icap]\n\t"
        "s_cbranch_scc1 PKB_NODEFER_%=\n\t"                 // chunk full: the packet keeps the node
        "s_lshl_b32 s89, %[cnt], 24\n\t"
        "s_or_b32 s89, s89, %[first]\n\t"
        "s_mov_b64 s[92:93], %[m]\n\t"
        "s_mov_b64 s[84:85], exec\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_mbcnt_lo_u32_b32 v24, s92, 0\n\t"
        "v_mbcnt_hi_u32_b32 v24, s93, v24\n\t"              // rank of the lane among the lanes of m
        "v_add_u32 v24, %[icnt], v24\n\t"
        "v_lshl_add_u32 v24, v24, 1, v24\n\t"               // x 3
        "v_lshlrev_b32 v24, 4, v24\n\t"                     // x 16: 48 bytes per item
        "v_mov_b32 v28, %[ox]\n\t"
        "v_mov_b32 v29, %[oy]\n\t"
        "v_mov_b32 v30, %[oz]\n\t"
        "v_mov_b32 v31, %[tcur]\n\t"
        "global_store_dwordx4 v24, v[28:31], %[ibase]\n\t"
        "s_nop 1\n\t"                                        // store data of more than 64 bits, then a VALU write of those registers: 1 wait state
        "v_mov_b32 v28, %[dx]\n\t"
        "v_mov_b32 v29, %[dy]\n\t"
        "v_mov_b32 v30, %[dz]\n\t"
        "v_mov_b32 v31, %[slot]\n\t"
        "global_store_dwordx4 v24, v[28:31], %[ibase] offset:16\n\t"
        "s_nop 1\n\t"
        "v_mov_b32 v28, s89\n\t"
        "v_mov_b32 v29, %[inst]\n\t"
        "global_store_dwordx2 v24, v[28:29], %[ibase] offset:32\n\t"
        "s_bcnt1_i32_b64 s88, s[92:93]\n\t"
        "s_add_u32 %[icnt], %[icnt], s88\n\t"
        "s_mov_b64 exec, s[84:85]\n\t"
        "s_branch PKB_POP_%=\n\t"
        "PKB_DEFER_LDS_%=:\n\t"
        "v_cmp_le_u32 vcc, 8, %[psp]\n\t"                   // every lane of m needs room on its private stack
        "s_and_b64 s[84:85], vcc, %[m]\n\t"
        "s_cmp_lg_u64 s[84:85], 0\n\t"
        "s_cbranch_scc1 PKB_NODEFER_%=\n\t"
        "s_lshl_b32 s89, %[cnt], 24\n\t"
        "s_or_b32 s89, s89, %[first]\n\t"
        "s_mov_b64 s[84:85], exec\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_lshl_add_u32 %[t0], %[psp], 8, %[llds]\n\t"
        "v_mov_b32 %[t1], s89\n\t"
        "ds_write_b32 %[t0], %[t1]\n\t"
        "v_add_u32 %[psp], 1, %[psp]\n\t"
        "s_mov_b64 exec, s[84:85]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_branch PKB_POP_%=\n\t"
        // ---------------- leaf: triangles first .. first + count - 1 ----------------
        "PKB_LEAF_%=:\n\t"
        "s_mov_b32 s98, %[first]\n\t"
        "s_add_i32 s99, %[first], %[cnt]\n\t"
        PK_ASM_TRI_LOOP
        // ---------------- pop ----------------
        "PKB_POP_%=:\n\t"
        "s_mov_b64 %[m], 0\n\t"
        "s_cmp_eq_u64 %[alive], 0\n\t"
        "s_cbranch_scc1 PKB_END_%=\n\t"
        "s_cmp_le_i32 %[sp], %[floor]\n\t"
        "s_cbranch_scc1 PKB_END_%=\n\t"
        "s_sub_i32 %[sp], %[sp], 1\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "v_readlane_b32 s84, %[stmlo], m0\n\t"
        "v_readlane_b32 s85, %[stmhi], m0\n\t"
        "v_readlane_b32 %[first], %[sta], m0\n\t"
        "v_readlane_b32 %[cnt], %[stb], m0\n\t"
        "s_and_b64 %[m], s[84:85], %[alive]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKB_POP_%=\n\t"
        "s_branch PKB_DISPATCH_%=\n\t"
        "PKB_END_%=:\n\t"
        : [first] "+&s"(W.cur_first), [cnt] "+&s"(W.cur_cnt), [m] "+&s"(W.m), [alive] "+&s"(W.alive), [sp] "+&s"(W.sp),
          [err] "+&s"(err), [steps] "+&s"(steps), [icnt] "+&s"(icnt),
          [sta] "+&v"(W.st_a), [stb] "+&v"(W.st_b), [stmlo] "+&v"(W.st_mlo), [stmhi] "+&v"(W.st_mhi), [psp] "+&v"(W.psp),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7),
          [t8] "=&v"(t8), [t9] "=&v"(t9)
        : [nodes] "s"(nodes), [tris] "s"(tris), [floor] "s"(floor_sp),
          [oxy] "v"(W.ray.o_xy), [ozz] "v"(W.ray.o_zz), [ixy] "v"(W.ray.i_xy), [izz] "v"(W.ray.i_zz), [tcur] "v"(W.ray.tcur),
          [ox] "v"(W.ray.o_xy.x), [oy] "v"(W.ray.o_xy.y), [oz] "v"(W.ray.o_zz.x), [dx] "v"(W.ray.d.x), [dy] "v"(W.ray.d.y), [dz] "v"(W.ray.d.z),
          [eps] "s"(eps), [t0d] "s"(defer_t0), [tld] "s"(defer_leaf), [gsh] "s"(grow_shift), [llds] "v"(lane_lds),
          [icap] "s"(icap), [ibase] "s"(item_base), [inst] "s"(inst), [slot] "v"(slot), [sgn] "s"(sgn)
        : "vcc", "scc", "m0", "memory",
          "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63",
          "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    W.pk_steps = steps; W.item_n = icnt;
    if (err) W.err = 1;
}

struct PkHybrid { int * stk; float * stk_key; uint32_t lane_lds; int * spill; float * spill_key; int spill_stride; int2 * fifo; int fifo_stride; int t0, t_leaf, grow; uint32_t item_cap; unsigned long long item_base; uint32_t gthread; int closest_asm; };

// ---- hybrid any-hit walk, per-lane phase -----------------------------------------------------------------------------------------
// Walks the subtrees the packet walk handed to the lanes' private stacks (entries are nodes that already passed the lane's slab test,
// packed like k_trace_fast's: 0x40000000 | axis << 28 | leaf count << 24 | first).  One ray per lane, per-lane gathers through the
// vector memory path, private stack striped through LDS ([entry][lane], RTX_PK_PRIV entries, deeper ones in HBM): the part of the tree
// where the rays of a packet no longer agree is walked the way the per-lane kernel walks it, the shared top of the tree was not.
#define RTX_PK_PRIV 24
RTX_D void pk_lane_phase_any(PkWalk & W, const int lane, int * const stk, int * const spill, const int spill_stride, rtx_gptr nodes, rtx_gptr tris) {
    bool live = (W.alive >> lane) & 1ull;
    int psp = live ? W.psp : 0;
    const v3 co = V3(W.ray.o_xy.x, W.ray.o_xy.y, W.ray.o_zz.x), cd = W.ray.d, cinv = V3(W.ray.i_xy.x, W.ray.i_xy.y, W.ray.i_zz.x);
    const float tmax = W.ray.tcur;
    bool have_cur = false; int cur_first = 0, cur_cnt = 0;
    auto push = [&](int e) {
        if (psp < RTX_PK_PRIV) stk[psp * RTX_WAVE] = e;
        else if (psp < RTX_MAX_STACK + 8) spill[(psp - RTX_PK_PRIV) * spill_stride] = e;
        else { W.err = 1; return; }
        psp++;
    };
#ifdef RTX_PK_PROF
    unsigned long long lane_iters = 0;
#endif
    for (;;) {
        if (__ballot(live && (have_cur || psp > 0)) == 0ull) break;
#ifdef RTX_PK_PROF
        lane_iters++;
#endif
        // inner node: child-pair gather, both slab tests, descend left, push right (any order is exact for any-hit rays)
        if (live && have_cur && (cur_cnt & 0x3fffffff) == 0) {
            const int left = cur_first;
            const float4 la = gld(nodes, 2 * left), lb = gld(nodes, 2 * left + 1), ra = gld(nodes, 2 * left + 2), rb = gld(nodes, 2 * left + 3);
            float k;
            const bool pl = slab_test_sel<true>(la, lb, co, cinv, tmax, k), pr = slab_test_sel<true>(ra, rb, co, cinv, tmax, k);
            if (pl) {
                if (pr) {
                    const int rf = as_i(ra.w), rc = as_i(rb.w), rl = rc & 0x3fffffff;
                    if ((uint32_t)rf < (1u << 24) && rl < 16) push((int)(0x40000000u | (((uint32_t)rc >> 30) << 28) | ((uint32_t)rl << 24) | (uint32_t)rf));
                    else push(left + 1);
                }
                cur_first = as_i(la.w); cur_cnt = as_i(lb.w);
            } else if (pr) { cur_first = as_i(ra.w); cur_cnt = as_i(rb.w); }
            else have_cur = false;
        }
        // leaf
        if (live && have_cur && (cur_cnt & 0x3fffffff) > 0) {
            have_cur = false;
            const int cnt = cur_cnt & 0x3fffffff;
            for (int i = cur_first; i < cur_first + cnt; i++) {
                const float4 p0 = gld(tris, RTX_TRI_STRIDE * i), e1 = gld(tris, RTX_TRI_STRIDE * i + 1), e2 = gld(tris, RTX_TRI_STRIDE * i + 2);
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
                if (!((t > RAY_EPSILON) & (t < tmax))) continue;
                live = false; break;
            }
        }
        // pop
        if (live && !have_cur && psp > 0) {
            psp--;
            const int e = (psp < RTX_PK_PRIV) ? stk[psp * RTX_WAVE] : spill[(psp - RTX_PK_PRIV) * spill_stride];
            if (e & 0x40000000) { cur_first = e & 0x00ffffff; cur_cnt = (int)((((uint32_t)e >> 28) & 3u) << 30) | ((e >> 24) & 15); }
            else { const float4 na = gld(nodes, 2 * e), nb = gld(nodes, 2 * e + 1); cur_first = as_i(na.w); cur_cnt = as_i(nb.w); }
            have_cur = true;
        }
    }
    W.alive &= __ballot(live);            // lanes that found an occluder leave the packet
    W.psp = 0;
#ifdef RTX_PK_PROF
    if (lane_iters > W.prof[9]) W.prof[9] = lane_iters;       // slot 9: most iterations of one per-lane phase
    if ((unsigned long long)W.pk_steps > W.prof[5]) W.prof[5] = (unsigned long long)W.pk_steps;   // slot 5: most packet steps of one walk
#endif
}

// The per-lane phase over the 4-wide records (entries: leaf count << 24 | first, nodes that passed the lane's slab test): one 128-byte
// gather feeds four slab tests, the lowest passing slot is walked next, the others go to the private stack.
RTX_D void pk_lane_phase_any4(PkWalk & W, const int lane, const uint32_t lane_lds, int * const spill, const int spill_stride, rtx_gptr nodes4, rtx_gptr tris) {
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int * const stk = (lds_int *)(uintptr_t)lane_lds;           // ds_write / ds_read, not flat accesses
    bool live = (W.alive >> lane) & 1ull;
    int psp = live ? W.psp : 0;
    const v3 co = V3(W.ray.o_xy.x, W.ray.o_xy.y, W.ray.o_zz.x), cd = W.ray.d;
    const float tmax = W.ray.tcur;
    bool have_cur = false; int cur_first = 0, cur_cnt = 0;
    auto push = [&](int e) {
        if (psp < RTX_PK_PRIV) stk[psp * RTX_WAVE] = e;
        else if (psp < RTX_MAX_STACK + 8) spill[(psp - RTX_PK_PRIV) * spill_stride] = e;
        else { W.err = 1; return; }
        psp++;
    };
    for (;;) {
        if (__ballot(live && (have_cur || psp > 0)) == 0ull) break;
        if (live && have_cur && cur_cnt == 0) {
            // the whole record first (one round trip), the four tests, then the bookkeeping
            const int base = 4 * cur_first;
            const float4 a0 = gld(nodes4, base), a1 = gld(nodes4, base + 1), b0 = gld(nodes4, base + 2), b1 = gld(nodes4, base + 3);
            const float4 c0 = gld(nodes4, base + 4), c1 = gld(nodes4, base + 5), d0 = gld(nodes4, base + 6), d1 = gld(nodes4, base + 7);
            const bool p0 = pk_slab<true>(a0, a1, W.ray), p1 = pk_slab<true>(b0, b1, W.ray), p2 = pk_slab<true>(c0, c1, W.ray), p3 = pk_slab<true>(d0, d1, W.ray);
            const int e1 = (int)(((uint32_t)as_i(b1.w) << 24) | (uint32_t)as_i(b1.z)), e2 = (int)(((uint32_t)as_i(c1.w) << 24) | (uint32_t)as_i(c1.z));
            const int e3 = (int)(((uint32_t)as_i(d1.w) << 24) | (uint32_t)as_i(d1.z));
            // slot k is pushed when it passes and a lower slot passes too (that one is walked first)
            const bool q3 = p3 & (p0 | p1 | p2), q2 = p2 & (p0 | p1), q1 = p1 & p0;
            if (psp + 3 <= RTX_PK_PRIV) {
                if (q3) { stk[psp * RTX_WAVE] = e3; psp++; }
                if (q2) { stk[psp * RTX_WAVE] = e2; psp++; }
                if (q1) { stk[psp * RTX_WAVE] = e1; psp++; }
            } else { if (q3) push(e3); if (q2) push(e2); if (q1) push(e1); }
            have_cur = p0 | p1 | p2 | p3;
            cur_first = p0 ? as_i(a1.z) : (p1 ? as_i(b1.z) : (p2 ? as_i(c1.z) : as_i(d1.z)));
            cur_cnt   = p0 ? as_i(a1.w) : (p1 ? as_i(b1.w) : (p2 ? as_i(c1.w) : as_i(d1.w)));
        }
        if (live && have_cur && cur_cnt > 0) {
            have_cur = false;
            for (int i = cur_first; i < cur_first + cur_cnt; i++) {
                const float4 p0 = gld(tris, RTX_TRI_STRIDE * i), e1 = gld(tris, RTX_TRI_STRIDE * i + 1), e2 = gld(tris, RTX_TRI_STRIDE * i + 2);
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
                if (!((t > RAY_EPSILON) & (t < tmax))) continue;
                live = false; break;
            }
        }
        if (live && !have_cur && psp > 0) {
            psp--;
            const int e = (psp < RTX_PK_PRIV) ? stk[psp * RTX_WAVE] : spill[(psp - RTX_PK_PRIV) * spill_stride];
            cur_first = e & 0x00ffffff; cur_cnt = (int)((uint32_t)e >> 24);
            have_cur = true;
        }
    }
    W.alive &= __ballot(live);
    W.psp = 0;
}

// ---- hybrid closest-hit walk, per-lane phase ----------------------------------------------------------------------------------
// A lane whose rays no longer agree with its packet turns PRIVATE: from then on every node the packet walk would have handed it (the
// node at which it left, then every shared stack entry carrying its bit, in the order they are popped = the lane's own visiting order)
// is appended to the lane's work list instead, and the lane takes no further part in the packet.  Here each lane walks its list front
// to back, every entry's subtree depth-first with its own stack — BottomLevelBVH::trace (BottomLevelBVH.cpp:355-396) for that ray from
// that point on: the far child re-tested at pop time against the then-current closest distance (entries not tested yet: the same test
// on the node's own box), the lane's own near / far order, strict comparisons.  Results are those of the reference's per-ray loop.
#define RTX_PK_FIFO 48                      // work-list entries per lane (HBM, [entry][thread]); a lane turns private only if the list cannot overflow
// A work-list entry is (reference, key).  reference: 0x40000000 | axis << 28 | leaf count << 24 | first (the node's own (first, count): nothing to
// fetch before its children / triangles; nodes with first >= 2^24 or 16+ triangles: the node index, fetched when its turn comes).  key: the
// lane's t_near of the node's box, computed by the shared walk when it popped the entry — the reference's pop-time test at the entry's turn is
// exactly `key < closest distance then` (slab_test_key); -inf for a node the lane has passed already (hand-over, parked sign split); NaN =
// "test the fetched box" (entries written by the compiled walk, which has no per-lane t_near at hand).
#define PK_FIFO_PACKED 0x40000000
#define PK_KEY_PASSED (-INFINITY)
// WIDE: the inner step reads the ordered 4-wide record of the node (build_nodes_pk4c, rtx_api.hip: the four grandchildren in the tree's shape),
// tests the four boxes and takes them in the order the two binary levels would — half the dependent fetches, and the two children's own tests
// fall away (nested boxes: a grandchild that passes at its pop has a parent that would have passed at its own).
template <bool WIDE>
RTX_D void pk_lane_phase_closest(PkWalk & W, const int lane, const bool ordered, const PkHybrid & H, rtx_gptr nodes, rtx_gptr nodes4, rtx_gptr tris) {
    const int fifo_n = W.fifo_n;
    int fi = 0, sp = 0;
    int2 nxt = fifo_n > 0 ? H.fifo[0] : make_int2(0, 0);          // the next work-list entry is always in flight before it is needed
    const v3 co = V3(W.ray.o_xy.x, W.ray.o_xy.y, W.ray.o_zz.x), cd = W.ray.d, cinv = V3(W.ray.i_xy.x, W.ray.i_xy.y, W.ray.i_zz.x);
    float tcur = W.ray.tcur;
    bool have_cur = false; int cur_first = 0, cur_cnt = 0;
    // BVHNode::should_visit_left_first per axis code (0: none / leaf, 1 x, 2 y, 3 z) as a bit set; BVH_TRAVERSE_TREE_NAIVE: always the left child first
    const uint32_t pos_bits = ordered ? (1u | (cd.x > 0.0f ? 2u : 0u) | (cd.y > 0.0f ? 4u : 0u) | (cd.z > 0.0f ? 8u : 0u)) : 15u;
    auto push = [&](int e, float key) {
        if (sp < RTX_LDS_STACK) { H.stk[sp * RTX_WAVE] = e; H.stk_key[sp * RTX_WAVE] = key; }
        else if (sp < RTX_MAX_STACK) { H.spill[(sp - RTX_LDS_STACK) * H.spill_stride] = e; H.spill_key[(sp - RTX_LDS_STACK) * H.spill_stride] = key; }
        else { W.err = 1; return; }
        sp++;
    };
#ifdef RTX_LANE_PROF
#define LP_NOW() ({ asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_memtime(); })
    const unsigned long long lp_t0 = LP_NOW(); unsigned long long lp_it = 0;
#endif
    for (;;) {
        if (__ballot(have_cur || sp > 0 || fi < fifo_n) == 0ull) break;
#ifdef RTX_LANE_PROF
        lp_it++;
        W.lp[6] = (uint32_t)rfl((int)(W.lp[6] + (uint32_t)__popcll(__ballot(have_cur || sp > 0 || fi < fifo_n))));      // lanes still walking in this iteration
        const bool lp_inner = __ballot(have_cur && (cur_cnt & 0x3fffffff) == 0) != 0ull;
        const unsigned long long lp_a = LP_NOW();
#endif
        // inner node: child-pair gather, both slab tests, the lane's own near / far order (BVHNode.h:30-40)
        if (WIDE && have_cur && (cur_cnt & 0x3fffffff) == 0) {
            const int base = 4 * cur_first;
            const float4 a0 = gld(nodes4, base), a1 = gld(nodes4, base + 1), b0 = gld(nodes4, base + 2), b1 = gld(nodes4, base + 3);
            const float4 c0 = gld(nodes4, base + 4), c1 = gld(nodes4, base + 5), d0 = gld(nodes4, base + 6), d1 = gld(nodes4, base + 7);
            float k0, k1, k2, k3;
            // AABB::intersect on a slot (min.x, min.y, max.x, max.y) (min.z, max.z, -, -) in packed fp32 — the arithmetic of pk_slab<true>, the result of
            // slab_test_sel<true>: passes, and t_near as the key of the pop-time test
            auto slab = [&](const float4 & q0, const float4 & q1, float & key) {
                const rtx_f2v t0xy = (rtx_f2v{ q0.x, q0.y } - W.ray.o_xy) * W.ray.i_xy;
                const rtx_f2v t1xy = (rtx_f2v{ q0.z, q0.w } - W.ray.o_xy) * W.ray.i_xy;
                const rtx_f2v tz   = (rtx_f2v{ q1.x, q1.y } - W.ray.o_zz) * W.ray.i_zz;
                const float t_near = pk_max_s(RAY_EPSILON, pk_max3(pk_min(t0xy.x, t1xy.x), pk_min(t0xy.y, t1xy.y), pk_min(tz.x, tz.y)));
                const float t_far  = pk_min(tcur, pk_min3(pk_max(t0xy.x, t1xy.x), pk_max(t0xy.y, t1xy.y), pk_max(tz.x, tz.y)));
                key = t_near;
                return t_near < t_far;
            };
            const bool p0 = slab(a0, a1, k0), p1 = slab(b0, b1, k1), p2 = slab(c0, c1, k2), p3 = slab(d0, d1, k3);
            // a slot as a stack entry: its node's own (first, count, axis); a key of +inf marks a slot that failed (or is unused)
            auto entry = [](const float4 & lo, const float4 & hi) { const uint32_t m = (uint32_t)as_i(hi.w); return (int)(0x40000000u | ((m >> 30) << 28) | ((m & 15u) << 24) | (uint32_t)as_i(hi.z)); };
            int e0 = entry(a0, a1), e1 = entry(b0, b1), e2 = entry(c0, c1), e3 = entry(d0, d1);
            k0 = p0 ? k0 : INFINITY; k1 = p1 ? k1 : INFINITY; k2 = p2 ? k2 : INFINITY; k3 = p3 ? k3 : INFINITY;
            // the lane's own near / far order at the node and at its two children (BVHNode.h:30-40); axis 0 = a leaf child: one slot, no order
            const bool lf_x = ((pos_bits >> ((uint32_t)cur_cnt >> 30)) & 1u) != 0u;
            const bool lf_l = ((pos_bits >> (((uint32_t)as_i(a1.w) >> 26) & 3u)) & 1u) != 0u, lf_r = ((pos_bits >> (((uint32_t)as_i(c1.w) >> 26) & 3u)) & 1u) != 0u;
            auto swap_if = [](const bool c, int & ea, float & ka, int & eb, float & kb) { const int te = c ? eb : ea; const float tk = c ? kb : ka; eb = c ? ea : eb; kb = c ? ka : kb; ea = te; ka = tk; };
            swap_if(!lf_l, e0, k0, e1, k1);
            swap_if(!lf_r, e2, k2, e3, k3);
            swap_if(!lf_x, e0, k0, e2, k2);
            swap_if(!lf_x, e1, k1, e3, k3);
            // visiting order e0 .. e3: the first that passed is walked now, the others wait — the last one deepest — for the reference's pop-time test
            const bool q0 = k0 < INFINITY, q1 = k1 < INFINITY, q2 = k2 < INFINITY, q3 = k3 < INFINITY;
            if (q3 && (q0 | q1 | q2)) push(e3, k3);
            if (q2 && (q0 | q1)) push(e2, k2);
            if (q1 && q0) push(e1, k1);
            const int e = q0 ? e0 : (q1 ? e1 : (q2 ? e2 : e3));
            have_cur = q0 | q1 | q2 | q3;
            cur_first = e & 0x00ffffff; cur_cnt = (int)((((uint32_t)e >> 28) & 3u) << 30) | ((e >> 24) & 15);
        }
        if (!WIDE && have_cur && (cur_cnt & 0x3fffffff) == 0) {
            const int left = cur_first;
            const float4 la = gld(nodes, 2 * left), lb = gld(nodes, 2 * left + 1), ra = gld(nodes, 2 * left + 2), rb = gld(nodes, 2 * left + 3);
            float kl = 0.0f, kr = 0.0f;
            const bool pl = slab_test_sel<true>(la, lb, co, cinv, tcur, kl), pr = slab_test_sel<true>(ra, rb, co, cinv, tcur, kr);
            bool left_first = true;
            if (ordered) {
                const int axis = (int)((uint32_t)cur_cnt >> 30);
                if (axis == 1) left_first = cd.x > 0.0f; else if (axis == 2) left_first = cd.y > 0.0f; else if (axis == 3) left_first = cd.z > 0.0f;
            }
            const bool p_near = left_first ? pl : pr, p_far = left_first ? pr : pl;
            const int near_first = left_first ? as_i(la.w) : as_i(ra.w), near_cnt = left_first ? as_i(lb.w) : as_i(rb.w);
            const int far_first  = left_first ? as_i(ra.w) : as_i(la.w), far_cnt  = left_first ? as_i(rb.w) : as_i(lb.w);
            if (p_near) {
                if (p_far) {
                    const int far_leaf = far_cnt & 0x3fffffff;
                    const float far_key = left_first ? kr : kl;
                    if ((uint32_t)far_first < (1u << 24) && far_leaf < 16)
                        push((int)(0x40000000u | (((uint32_t)far_cnt >> 30) << 28) | ((uint32_t)far_leaf << 24) | (uint32_t)far_first), far_key);
                    else push(left_first ? left + 1 : left, far_key);
                }
                cur_first = near_first; cur_cnt = near_cnt;
            } else if (p_far) { cur_first = far_first; cur_cnt = far_cnt; }
            else have_cur = false;
        }
#ifdef RTX_LANE_PROF
        const unsigned long long lp_c = LP_NOW();                  // outside the divergent block: wave-uniform bookkeeping
        if (lp_inner) { W.lp[2] = (uint32_t)rfl((int)(W.lp[2] + 1u)); W.lp[7] = (uint32_t)rfl((int)(W.lp[7] + (uint32_t)(lp_c - lp_a))); }       // the whole inner block: child-pair gather + two slab tests + push
        const bool lp_leaf = __ballot(have_cur && (cur_cnt & 0x3fffffff) > 0) != 0ull;
        if (lp_leaf) W.lp[3] = (uint32_t)rfl((int)(W.lp[3] + 1u));
#endif
        // leaf: triangle_trace, BottomLevelBVH.cpp:214-309 (hit test; the accept branch is rebuilt by k_shade)
        if (have_cur && (cur_cnt & 0x3fffffff) > 0) {
            have_cur = false;
            const int end = cur_first + (cur_cnt & 0x3fffffff);
            // Two triangles per round trip: the reference's SBVH leaves hold two triangles (93 % of the cfg3 tree's), and a lane's walk is a chain
            // of dependent fetches — the second triangle's record (the next 64 bytes) travels with the first's.  They are still TESTED one after
            // the other, the second against the closest distance the first may have just set: the reference's loop, BottomLevelBVH.cpp:376-379.
            for (int i = cur_first; i < end; i += 2) {
                const bool two = i + 1 < end;
                const int j = two ? i + 1 : i;
                const float3 p0 = gld3(tris, RTX_TRI_STRIDE * i), e1 = gld3(tris, RTX_TRI_STRIDE * i + 1), e2 = gld3(tris, RTX_TRI_STRIDE * i + 2);
                const float3 q0 = gld3(tris, RTX_TRI_STRIDE * j), f1 = gld3(tris, RTX_TRI_STRIDE * j + 1), f2 = gld3(tris, RTX_TRI_STRIDE * j + 2);
                __builtin_amdgcn_sched_barrier(0);          // all six loads are issued before the first use (the scheduler had put the first test between the two triangles' loads)
                auto test = [&](const float3 & a0, const float3 & a1, const float3 & a2, const int idx) {
                    const v3 edge_1 = V3(a1.x, a1.y, a1.z), edge_2 = V3(a2.x, a2.y, a2.z);
                    const v3 h = vcross(cd, edge_2);
                    const float a = vdot(edge_1, h);
                    const float f = 1.0f / a;
                    const v3 s = vsub(co, V3(a0.x, a0.y, a0.z));
                    const float u = f * vdot(s, h);
                    if (!((u > 0.0f) & (u < 1.0f))) return;
                    const v3 qq = vcross(s, edge_1);
                    const float v = f * vdot(cd, qq);
                    if (!((v > 0.0f) & ((u + v) < 1.0f))) return;
                    const float t = f * vdot(edge_2, qq);
                    if (!((t > RAY_EPSILON) & (t < tcur))) return;
                    tcur = t; W.best_u = u; W.best_v = v; W.best_tri = idx; W.best_prim = PRIM_TRI | W.cur_inst;
                };
                test(p0, e1, e2, i);
                if (two) test(q0, f1, f2, j);
            }
        }
#ifdef RTX_LANE_PROF
        const unsigned long long lp_d = LP_NOW();
        if (lp_leaf) W.lp[8] = (uint32_t)rfl((int)(W.lp[8] + (uint32_t)(lp_d - lp_c)));
        const bool lp_popg = __ballot(!have_cur && sp > 0) != 0ull, lp_fifo = __ballot(!have_cur && sp == 0 && fi < fifo_n) != 0ull;
        if (lp_popg) W.lp[4] = (uint32_t)rfl((int)(W.lp[4] + 1u));
        if (lp_fifo) W.lp[5] = (uint32_t)rfl((int)(W.lp[5] + 1u));
#endif
        // pop: the lane's own stack first, then the next entry of its work list
        if (!have_cur) {
            if (sp > 0) {
                sp--;
                float key; int e;
                if (sp < RTX_LDS_STACK) { e = H.stk[sp * RTX_WAVE]; key = H.stk_key[sp * RTX_WAVE]; }
                else { e = H.spill[(sp - RTX_LDS_STACK) * H.spill_stride]; key = H.spill_key[(sp - RTX_LDS_STACK) * H.spill_stride]; }
                if (key < tcur) {                       // the reference's pop-time test (slab_test_key: t_near < closest distance now)
                    if (e & 0x40000000) { cur_first = e & 0x00ffffff; cur_cnt = (int)((((uint32_t)e >> 28) & 3u) << 30) | ((e >> 24) & 15); }
                    else { const float4 na = gld(nodes, 2 * e), nb = gld(nodes, 2 * e + 1); cur_first = as_i(na.w); cur_cnt = as_i(nb.w); }
                    have_cur = true;
                }
            } else if (fi < fifo_n) {
                const int e = nxt.x; const float key = as_f(nxt.y);
                fi++;
                if (fi < fifo_n) nxt = H.fifo[fi * H.fifo_stride];
                if (e & PK_FIFO_PACKED) {
                    if (key < tcur) { cur_first = e & 0x00ffffff; cur_cnt = (int)((((uint32_t)e >> 28) & 3u) << 30) | ((e >> 24) & 15); have_cur = true; }
                } else {
                    const float4 na = gld(nodes, 2 * e), nb = gld(nodes, 2 * e + 1);
                    float k;
                    const bool pass = (key != key) ? slab_test_sel<true>(na, nb, co, cinv, tcur, k) : (key < tcur);
                    if (pass) { cur_first = as_i(na.w); cur_cnt = as_i(nb.w); have_cur = true; }
                }
            }
        }
#ifdef RTX_LANE_PROF
        { const unsigned long long lp_f = LP_NOW(); if (lp_popg || lp_fifo) W.lp[9] = (uint32_t)rfl((int)(W.lp[9] + (uint32_t)(lp_f - lp_d))); }
#endif
    }
#ifdef RTX_LANE_PROF
    {
        const unsigned long long lp_e = LP_NOW();
        W.lp[0] = (uint32_t)rfl((int)(W.lp[0] + (uint32_t)(lp_e - lp_t0))); W.lp[1] = (uint32_t)rfl((int)(W.lp[1] + (uint32_t)(lp_it))); W.lp[12] = (uint32_t)rfl((int)(W.lp[12] + 1u));
        if ((uint32_t)(lp_e - lp_t0) > W.lp[10]) W.lp[10] = (uint32_t)rfl((int)(uint32_t)(lp_e - lp_t0));
        if ((uint32_t)lp_it > W.lp[11]) W.lp[11] = (uint32_t)rfl((int)(uint32_t)lp_it);
    }
#endif
    W.ray.tcur = tcur;
    W.fifo_n = 0;
}

// ---- the whole closest-hit walk of one BLAS for the lanes that walk TOGETHER, hand-scheduled -------------------------------------------
// BottomLevelBVH::trace (BottomLevelBVH.cpp:355-396) for the lanes of W.m from the node in W.cur_*, until the packet stack is back at
// W.floor_sp: ordered descend with sign split (the asm of pk_descend), leaves (triangle_trace's hit test with the triangle in SGPRs,
// every lane against its OWN closest distance, updated under the hit mask), pops with the reference's pop-time re-test of a far child
// (its own box fetched through the scalar cache, 32 bytes), and the hybrid hand-over: a node wanted by few lanes, and every later
// stack entry carrying a private lane's bit, goes to those lanes' work lists (HBM, [entry][thread]) for pk_lane_phase_closest.
// Exactness is the argument at the top of this file: every lane meets the nodes the reference pops and passes for it, in its own
// near / far order (a far child is pushed with the lanes that pass NOW and re-tested at pop time against their distances THEN).
//   stack entry: a = first, b = count | axis << 30, mask, idx = node index, df = depth | PK_FLAG_PRETESTED
//   SGPRs: s[64:79] child pair / triangle / popped node, s[80:93] temporaries, s[94:97] bases, s98 / s99 triangle loop; VGPRs v[24:31] + operands.
#define PK_ASM_REF /* s79 = work-list reference of the node in (first, cnt, idx): its packed (first, count) or, where that does not fit, its index */ \
        "s_and_b32 s76, %[cnt], 0x3fffffff\n\t"                                                                                     \
        "s_cmp_lt_u32 s76, 16\n\t"                                                                                                  \
        "s_cselect_b32 s77, 1, 0\n\t"                                                                                               \
        "s_cmp_lt_u32 %[first], 0x1000000\n\t"                                                                                      \
        "s_cselect_b32 s78, 1, 0\n\t"                                                                                               \
        "s_and_b32 s77, s77, s78\n\t"                                                                                               \
        "s_lshr_b32 s78, %[cnt], 30\n\t"                                                                                            \
        "s_lshl_b32 s78, s78, 28\n\t"                                                                                               \
        "s_lshl_b32 s76, s76, 24\n\t"                                                                                               \
        "s_or_b32 s78, s78, s76\n\t"                                                                                                \
        "s_or_b32 s78, s78, %[first]\n\t"                                                                                           \
        "s_or_b32 s78, s78, 0x40000000\n\t"                                                                                         \
        "s_cmp_lg_u32 s77, 0\n\t"                                                                                                   \
        "s_cselect_b32 s79, s78, %[idx]\n\t"
#define PK_ASM_FIFO_APPEND(MASK_LO, MASK_HI, KEY) /* (s79, KEY) to the work lists of the lanes of the mask */                        \
        "s_mov_b64 s[92:93], exec\n\t"                                                                                              \
        "s_mov_b32 exec_lo, " MASK_LO "\n\t"                                                                                        \
        "s_mov_b32 exec_hi, " MASK_HI "\n\t"                                                                                        \
        "v_mov_b32 v28, s79\n\t"                                                                                                    \
        "v_mov_b32 v29, " KEY "\n\t"                                                                                                \
        "global_store_dwordx2 %[fvoff], v[28:29], %[fbase]\n\t"                                                                     \
        "v_add_u32 %[fvoff], %[fstride], %[fvoff]\n\t"                                                                              \
        "v_add_u32 %[fn], 1, %[fn]\n\t"                                                                                             \
        "s_mov_b64 exec, s[92:93]\n\t"
RTX_D void pk_blas_closest_asm(PkWalk & W, const int stack_size, const int ordered, const PkHybrid & H, const int prim, const uint32_t gthread) {
    W.cur_first = rfl(W.cur_first); W.cur_cnt = rfl(W.cur_cnt); W.cur_idx = rfl(W.cur_idx); W.m = rfl64(W.m); W.sp = rfl(W.sp);
    W.depth = rfl(W.depth); W.pub = rfl64(W.pub);
    int err = 0, steps = rfl(W.pk_steps);
    const int floor_sp = rfl(W.floor_sp), ssz = rfl(stack_size), ord = rfl(ordered), t0d = rfl(H.t0), gsh = rfl(H.grow), primv = rfl(prim);
    const unsigned long long posx = __ballot(W.ray.d.x > 0.0f), posy = __ballot(W.ray.d.y > 0.0f), posz = __ballot(W.ray.d.z > 0.0f);
    const float eps = RAY_EPSILON;
    const unsigned long long nodes = rfl64((unsigned long long)W.nodes), tris = rfl64((unsigned long long)W.tris);
    const unsigned long long fbase = rfl64((unsigned long long)(H.fifo - gthread));          // H.fifo = work lists + this thread's index
    const int fstride = rfl(H.fifo_stride * 8);
    uint32_t fvoff = (gthread + (uint32_t)W.fifo_n * (uint32_t)H.fifo_stride) * 8u;           // byte offset of this lane's next work-list entry
    float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm volatile(
        "s_mov_b64 s[94:95], %[nodes]\n\t"
        "s_mov_b64 s[96:97], %[tris]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKC_POP_%=\n\t"
        "PKC_DISPATCH_%=:\n\t"
        // hybrid: a node wanted by few lanes (threshold grows with the length of the walk) is not walked by the packet
        "s_bcnt1_i32_b64 s89, %[m]\n\t"
        "s_lshr_b32 s90, %[steps], %[gsh]\n\t"
        "s_add_i32 s90, s90, %[t0d]\n\t"
        "s_add_i32 %[steps], %[steps], 1\n\t"
        "s_cmp_le_u32 s89, s90\n\t"
        "s_cbranch_scc0 PKC_WALK_%=\n\t"
        "s_sub_i32 s90, %[sp], %[floor]\n\t"                 // the lanes' work lists must be able to take every entry still on the stack
        "s_add_i32 s90, s90, 2\n\t"
        "s_cmp_le_i32 s90, %[fifomax]\n\t"
        "s_cbranch_scc0 PKC_WALK_%=\n\t"
        "s_mov_b64 s[86:87], %[m]\n\t"
        PK_ASM_REF
        PK_ASM_FIFO_APPEND("s86", "s87", "0xff800000")        // key -inf: the lanes have just passed this node
        "s_andn2_b64 %[pub], %[pub], %[m]\n\t"
        "s_branch PKC_POP_%=\n\t"
        "PKC_WALK_%=:\n\t"
        "s_and_b32 s88, %[cnt], 0x3fffffff\n\t"
        "s_cmp_lg_u32 s88, 0\n\t"
        "s_cbranch_scc1 PKC_LEAF_%=\n\t"
        // ---------------- inner node (pk_descend's step) ----------------
        "s_add_i32 s90, %[depth], 2\n\t"
        "s_cmp_gt_i32 s90, %[ssz]\n\t"
        "s_cselect_b32 %[err], 1, %[err]\n\t"
        "s_lshl_b32 s92, %[first], 5\n\t"
        "s_load_dwordx16 s[64:79], s[94:95], s92\n\t"
        // order: every lane must see ITS OWN near / far order; lanes that disagree are parked on this node
        "s_mov_b32 s91, 1\n\t"
        "s_cmp_eq_u32 %[ordered], 0\n\t"
        "s_cbranch_scc1 PKC_ORD_%=\n\t"
        "s_lshr_b32 s90, %[cnt], 30\n\t"
        "s_cmp_eq_u32 s90, 0\n\t"
        "s_cbranch_scc1 PKC_ORD_%=\n\t"
        "s_mov_b64 s[86:87], %[posx]\n\t"
        "s_cmp_eq_u32 s90, 2\n\t"
        "s_cselect_b64 s[86:87], %[posy], s[86:87]\n\t"
        "s_cmp_eq_u32 s90, 3\n\t"
        "s_cselect_b64 s[86:87], %[posz], s[86:87]\n\t"
        "s_and_b64 s[86:87], s[86:87], %[m]\n\t"
        "s_cmp_eq_u64 s[86:87], 0\n\t"
        "s_cselect_b32 s91, 0, 1\n\t"
        "s_cbranch_scc1 PKC_ORD_%=\n\t"
        "s_cmp_eq_u64 s[86:87], %[m]\n\t"
        "s_cbranch_scc1 PKC_ORD_%=\n\t"
        // split: park (node, m & ~pos) as a pretested entry, continue with the left-first lanes
        "s_andn2_b64 s[88:89], %[m], s[86:87]\n\t"
        "s_mov_b64 %[m], s[86:87]\n\t"
        "s_cmp_ge_i32 %[sp], 64\n\t"
        "s_cselect_b32 %[err], 1, %[err]\n\t"
        "s_cbranch_scc1 PKC_ORD_%=\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "s_or_b32 s90, %[depth], 0x20000000\n\t"
        "v_writelane_b32 %[sta], %[first], m0\n\t"
        "v_writelane_b32 %[stb], %[cnt], m0\n\t"
        "v_writelane_b32 %[stmlo], s88, m0\n\t"
        "v_writelane_b32 %[stmhi], s89, m0\n\t"
        "v_writelane_b32 %[stdf], s90, m0\n\t"
        "v_writelane_b32 %[stidx], %[idx], m0\n\t"
        "s_add_i32 %[sp], %[sp], 1\n\t"
        "PKC_ORD_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[82:83]")
        PK_ASM_SLAB(72, 73, 74, 75, 76, 77, "s[84:85]")
        // near / far (left child index = first, right = first + 1)
        "s_add_i32 s90, %[first], 1\n\t"
        "s_cmp_lg_u32 s91, 0\n\t"
        "s_cselect_b64 s[86:87], s[82:83], s[84:85]\n\t"
        "s_cselect_b64 s[88:89], s[84:85], s[82:83]\n\t"
        "s_cselect_b32 s80, s70, s78\n\t"
        "s_cselect_b32 s81, s71, s79\n\t"
        "s_cselect_b32 s92, s78, s70\n\t"
        "s_cselect_b32 s93, s79, s71\n\t"
        "s_cselect_b32 s82, %[first], s90\n\t"          // near index
        "s_cselect_b32 s83, s90, %[first]\n\t"          // far index
        "s_cmp_eq_u64 s[86:87], 0\n\t"
        "s_cbranch_scc1 PKC_NONEAR_%=\n\t"
        "s_cmp_eq_u64 s[88:89], 0\n\t"
        "s_cbranch_scc1 PKC_NOFAR_%=\n\t"
        "s_cmp_ge_i32 %[sp], 64\n\t"
        "s_cbranch_scc1 PKC_OVER_%=\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "s_nop 0\n\t"
        "v_writelane_b32 %[sta], s92, m0\n\t"
        "v_writelane_b32 %[stb], s93, m0\n\t"
        "v_writelane_b32 %[stmlo], s88, m0\n\t"
        "v_writelane_b32 %[stmhi], s89, m0\n\t"
        "v_writelane_b32 %[stdf], %[depth], m0\n\t"
        "v_writelane_b32 %[stidx], s83, m0\n\t"
        "s_add_i32 %[sp], %[sp], 1\n\t"
        "PKC_NOFAR_%=:\n\t"
        "s_mov_b32 %[first], s80\n\t"
        "s_mov_b32 %[cnt], s81\n\t"
        "s_mov_b32 %[idx], s82\n\t"
        "s_mov_b64 %[m], s[86:87]\n\t"
        "s_add_i32 %[depth], %[depth], 1\n\t"
        "s_branch PKC_DISPATCH_%=\n\t"
        "PKC_OVER_%=:\n\t"
        "s_mov_b32 %[err], 1\n\t"
        "s_branch PKC_NOFAR_%=\n\t"
        "PKC_NONEAR_%=:\n\t"
        "s_cmp_eq_u64 s[88:89], 0\n\t"
        "s_cbranch_scc1 PKC_POP_%=\n\t"
        "s_mov_b32 %[first], s92\n\t"
        "s_mov_b32 %[cnt], s93\n\t"
        "s_mov_b32 %[idx], s83\n\t"
        "s_mov_b64 %[m], s[88:89]\n\t"
        "s_branch PKC_DISPATCH_%=\n\t"
        // ---------------- leaf: triangle_trace's hit test, triangles first .. first + count - 1 (BottomLevelBVH.cpp:214-246) ----------------
        "PKC_LEAF_%=:\n\t"
        "s_mov_b32 s98, %[first]\n\t"
        "s_add_i32 s99, %[first], s88\n\t"
        "PKC_TRI_%=:\n\t"
        "s_lshl_b32 s89, s98, 6\n\t"
        "s_load_dwordx8 s[64:71], s[96:97], s89\n\t"
        "s_add_u32 s89, s89, 32\n\t"
        "s_load_dwordx4 s[72:75], s[96:97], s89\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        /* h = cross(d, e2)                                      p0 = s64..66, e1 = s68..70, e2 = s72..74 */
        "v_mul_f32 %[t0], s74, %[dy]\n\t"
        "v_mul_f32 %[t1], s73, %[dz]\n\t"
        "v_mul_f32 %[t2], s72, %[dz]\n\t"
        "v_mul_f32 %[t3], s74, %[dx]\n\t"
        "v_mul_f32 %[t4], s73, %[dx]\n\t"
        "v_mul_f32 %[t5], s72, %[dy]\n\t"
        "v_sub_f32 v24, %[t0], %[t1]\n\t"
        "v_sub_f32 v25, %[t2], %[t3]\n\t"
        "v_sub_f32 v26, %[t4], %[t5]\n\t"
        /* a = e1.x*h.x + (e1.y*h.y + e1.z*h.z) */
        "v_mul_f32 %[t0], s69, v25\n\t"
        "v_mul_f32 %[t1], s70, v26\n\t"
        "v_mul_f32 %[t2], s68, v24\n\t"
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_add_f32 %[t9], %[t2], %[t0]\n\t"
        /* f = 1.0f / a, correctly rounded */
        "v_div_scale_f32 %[t0], s[84:85], %[t9], %[t9], 1.0\n\t"
        "v_rcp_f32 %[t1], %[t0]\n\t"
        "v_div_scale_f32 %[t2], vcc, 1.0, %[t9], 1.0\n\t"
        "v_subrev_f32 v27, s64, %[ox]\n\t"
        "v_fma_f32 %[t3], -%[t0], %[t1], 1.0\n\t"
        "v_fmac_f32 %[t1], %[t3], %[t1]\n\t"
        "v_mul_f32 %[t4], %[t2], %[t1]\n\t"
        "v_fma_f32 %[t5], -%[t0], %[t4], %[t2]\n\t"
        "v_fmac_f32 %[t4], %[t5], %[t1]\n\t"
        "v_fma_f32 %[t0], -%[t0], %[t4], %[t2]\n\t"
        "v_subrev_f32 v28, s65, %[oy]\n\t"
        "v_div_fmas_f32 %[t0], %[t0], %[t1], %[t4]\n\t"
        "v_subrev_f32 v29, s66, %[oz]\n\t"
        "v_div_fixup_f32 %[t8], %[t0], %[t9], 1.0\n\t"       /* f */
        /* u = f * (s.x*h.x + (s.y*h.y + s.z*h.z)) */
        "v_mul_f32 %[t0], v28, v25\n\t"
        "v_mul_f32 %[t1], v29, v26\n\t"
        "v_mul_f32 %[t2], v27, v24\n\t"
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"
        "v_mul_f32 %[t7], %[t8], %[t0]\n\t"                  /* u */
        "v_cmp_lt_f32 s[84:85], 0, %[t7]\n\t"
        "v_cmp_gt_f32 s[86:87], 1.0, %[t7]\n\t"
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"
        "s_and_b64 s[84:85], s[84:85], %[m]\n\t"
        "s_cmp_lg_u64 s[84:85], 0\n\t"
        "s_cbranch_scc0 PKC_NEXT_%=\n\t"                     /* no lane of the node passes the u test */
        /* q = cross(s, e1) */
        "v_mul_f32 %[t0], s70, v28\n\t"
        "v_mul_f32 %[t1], s69, v29\n\t"
        "v_mul_f32 %[t2], s68, v29\n\t"
        "v_mul_f32 %[t3], s70, v27\n\t"
        "v_mul_f32 %[t4], s69, v27\n\t"
        "v_mul_f32 %[t5], s68, v28\n\t"
        "v_sub_f32 v24, %[t0], %[t1]\n\t"
        "v_sub_f32 v25, %[t2], %[t3]\n\t"
        "v_sub_f32 v26, %[t4], %[t5]\n\t"
        /* v = f * (d.x*q.x + (d.y*q.y + d.z*q.z)) */
        "v_mul_f32 %[t0], %[dy], v25\n\t"
        "v_mul_f32 %[t1], %[dz], v26\n\t"
        "v_mul_f32 %[t2], %[dx], v24\n\t"
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"
        "v_mul_f32 %[t6], %[t8], %[t0]\n\t"                  /* v */
        /* t = f * (e2.x*q.x + (e2.y*q.y + e2.z*q.z)) */
        "v_mul_f32 %[t0], s73, v25\n\t"
        "v_mul_f32 %[t1], s74, v26\n\t"
        "v_mul_f32 %[t2], s72, v24\n\t"
        "v_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_add_f32 %[t0], %[t2], %[t0]\n\t"
        "v_mul_f32 %[t5], %[t8], %[t0]\n\t"                  /* t */
        "v_add_f32 %[t4], %[t7], %[t6]\n\t"                  /* u + v */
        "v_cmp_lt_f32 s[86:87], 0, %[t6]\n\t"
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"
        "v_cmp_gt_f32 s[86:87], 1.0, %[t4]\n\t"
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"
        "v_cmp_lt_f32 s[86:87], %[eps], %[t5]\n\t"
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"
        "v_cmp_lt_f32 s[86:87], %[t5], %[tcur]\n\t"          /* t < this lane's closest distance so far */
        "s_and_b64 s[84:85], s[84:85], s[86:87]\n\t"
        "s_cmp_lg_u64 s[84:85], 0\n\t"
        "s_cbranch_scc0 PKC_NEXT_%=\n\t"
        /* the lanes that are hit take the triangle (BottomLevelBVH.cpp:248-250; the attributes are rebuilt when the hit is shaded) */
        "s_mov_b64 s[92:93], exec\n\t"
        "s_mov_b64 exec, s[84:85]\n\t"
        "v_mov_b32 %[tcur], %[t5]\n\t"
        "v_mov_b32 %[bu], %[t7]\n\t"
        "v_mov_b32 %[bv], %[t6]\n\t"
        "v_mov_b32 %[btri], s98\n\t"
        "v_mov_b32 %[bprim], %[prim]\n\t"
        "s_mov_b64 exec, s[92:93]\n\t"
        "PKC_NEXT_%=:\n\t"
        "s_add_i32 s98, s98, 1\n\t"
        "s_cmp_lt_i32 s98, s99\n\t"
        "s_cbranch_scc1 PKC_TRI_%=\n\t"
        // ---------------- pop ----------------
        "PKC_POP_%=:\n\t"
        "s_mov_b64 %[m], 0\n\t"
        "s_cmp_le_i32 %[sp], %[floor]\n\t"
        "s_cbranch_scc1 PKC_END_%=\n\t"
        "s_sub_i32 %[sp], %[sp], 1\n\t"
        "s_mov_b32 m0, %[sp]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 s84, %[stmlo], m0\n\t"
        "v_readlane_b32 s85, %[stmhi], m0\n\t"
        "v_readlane_b32 %[first], %[sta], m0\n\t"
        "v_readlane_b32 %[cnt], %[stb], m0\n\t"
        "v_readlane_b32 s91, %[stdf], m0\n\t"
        "v_readlane_b32 %[idx], %[stidx], m0\n\t"
        "s_and_b32 %[depth], s91, 0xffff\n\t"
        "s_bitcmp1_b32 s91, 29\n\t"                           // PK_FLAG_PRETESTED: lanes parked at a sign split have passed this node already
        "s_cbranch_scc0 PKC_FAR_%=\n\t"
        // parked split: its private lanes take it as it is (key -inf), its public lanes walk it now
        "s_andn2_b64 s[86:87], s[84:85], %[pub]\n\t"
        "s_cmp_lg_u64 s[86:87], 0\n\t"
        "s_cbranch_scc0 PKC_PRE_PUB_%=\n\t"
        PK_ASM_REF
        PK_ASM_FIFO_APPEND("s86", "s87", "0xff800000")
        "PKC_PRE_PUB_%=:\n\t"
        "s_and_b64 %[m], s[84:85], %[pub]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKC_POP_%=\n\t"
        "s_branch PKC_DISPATCH_%=\n\t"
        // far child: the reference tests it when it pops it, against the closest distance THEN (BottomLevelBVH.cpp:366-370).  Public lanes: now.
        // Private lanes: at the entry's turn in their own walk — the test is `t_near < distance then` (slab_test_key), so their t_near travels
        // with the entry as its key; lanes that fail against their distance NOW can only fail later (the distance never grows): dropped here.
        "PKC_FAR_%=:\n\t"
        "s_mov_b64 %[m], s[84:85]\n\t"
        "s_lshl_b32 s88, %[idx], 5\n\t"
        "s_load_dwordx8 s[64:71], s[94:95], s88\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        PK_ASM_SLAB(64, 65, 66, 67, 68, 69, "s[80:81]")
        "s_andn2_b64 s[86:87], s[80:81], %[pub]\n\t"
        "s_cmp_lg_u64 s[86:87], 0\n\t"
        "s_cbranch_scc0 PKC_FAR_PUB_%=\n\t"
        PK_ASM_REF
        PK_ASM_FIFO_APPEND("s86", "s87", "v30")              // v30 = this lane's t_near of the box (PK_ASM_SLAB)
        "PKC_FAR_PUB_%=:\n\t"
        "s_and_b64 %[m], s[80:81], %[pub]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 PKC_POP_%=\n\t"
        "s_branch PKC_DISPATCH_%=\n\t"
        "PKC_END_%=:\n\t"
        : [first] "+&s"(W.cur_first), [cnt] "+&s"(W.cur_cnt), [idx] "+&s"(W.cur_idx), [m] "+&s"(W.m), [sp] "+&s"(W.sp), [depth] "+&s"(W.depth),
          [err] "+&s"(err), [steps] "+&s"(steps), [pub] "+&s"(W.pub),
          [sta] "+&v"(W.st_a), [stb] "+&v"(W.st_b), [stmlo] "+&v"(W.st_mlo), [stmhi] "+&v"(W.st_mhi), [stdf] "+&v"(W.st_df), [stidx] "+&v"(W.st_idx),
          [tcur] "+&v"(W.ray.tcur), [bu] "+&v"(W.best_u), [bv] "+&v"(W.best_v), [btri] "+&v"(W.best_tri), [bprim] "+&v"(W.best_prim),
          [fvoff] "+&v"(fvoff), [fn] "+&v"(W.fifo_n),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7),
          [t8] "=&v"(t8), [t9] "=&v"(t9)
        : [nodes] "s"(nodes), [tris] "s"(tris), [floor] "s"(floor_sp), [ssz] "s"(ssz), [eps] "s"(eps), [ordered] "s"(ord), [t0d] "s"(t0d), [gsh] "s"(gsh),
          [posx] "s"(posx), [posy] "s"(posy), [posz] "s"(posz), [fbase] "s"(fbase), [fstride] "s"(fstride), [prim] "s"(primv), [fifomax] "s"((int)RTX_PK_FIFO),
          [oxy] "v"(W.ray.o_xy), [ozz] "v"(W.ray.o_zz), [ixy] "v"(W.ray.i_xy), [izz] "v"(W.ray.i_zz),
          [ox] "v"(W.ray.o_xy.x), [oy] "v"(W.ray.o_xy.y), [oz] "v"(W.ray.o_zz.x), [dx] "v"(W.ray.d.x), [dy] "v"(W.ray.d.y), [dz] "v"(W.ray.d.z)
        : "vcc", "scc", "m0", "memory",
          "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    W.pk_steps = steps;
    if (err) W.err = 1;
}

// Walks until the packet is finished (returns true) or, FAST only, until an instance is entered in which some lane's inverse
// direction is not finite (returns false: the caller continues with the reference-form variant; the state is consistent).
// `world(o, d)` reloads a lane's world-space ray (kept in memory, not in registers).
template <bool ANY, bool PSTAT, bool FAST, bool ASM, bool SPLIT, typename WorldFn>
RTX_D bool pk_walk(const DevScene & sc, PkWalk & W, const int lane, const bool ordered, const PkHybrid & H, WorldFn && world) {
    for (;;) {
        if (W.m != 0ull) {
            if (!ANY && ASM && FAST && W.floor_sp >= 0) {
                // closest-hit hybrid: a node wanted by few lanes (threshold grows with the length of the walk) is not walked by the packet;
                // its lanes turn private and take the node (which they have just passed) as the first entry of their work lists
                const int thr = H.t0 + (W.pk_steps >> H.grow);
                if (__popcll(W.m) <= thr && (W.sp - W.floor_sp) + 2 <= RTX_PK_FIFO) {
                    if ((W.m >> lane) & 1ull) { H.fifo[W.fifo_n * H.fifo_stride] = make_int2(W.cur_idx, as_i(PK_KEY_PASSED)); W.fifo_n++; }
                    W.pub &= ~W.m; W.m = 0ull;
                    continue;
                }
                W.pk_steps++;
            }
            const int leafc = W.cur_cnt & 0x3fffffff;
            if (leafc > 0) {
                PK_T0();
                if (W.floor_sp >= 0) {
                    // BLAS leaf: triangle_trace / triangle_intersect (BottomLevelBVH.cpp:214-346) with the triangle in SGPRs
                    const v3 co = V3(W.ray.o_xy.x, W.ray.o_xy.y, W.ray.o_zz.x), cd = W.ray.d;
                    for (int i = W.cur_first; i < W.cur_first + leafc; i++) {
                        const float4 p0 = sld(W.tris, RTX_TRI_STRIDE * i), e1 = sld(W.tris, RTX_TRI_STRIDE * i + 1), e2 = sld(W.tris, RTX_TRI_STRIDE * i + 2);
                        if (PSTAT) { W.st_tris++; W.st_tri_lanes += (unsigned long long)__popcll(W.m); }
                        const v3 edge_1 = V3(e1.x, e1.y, e1.z), edge_2 = V3(e2.x, e2.y, e2.z);
                        const v3 h = vcross(cd, edge_2);
                        const float a = vdot(edge_1, h);
                        const float f = 1.0f / a;
                        const v3 s = vsub(co, V3(p0.x, p0.y, p0.z));
                        const float u = f * vdot(s, h);
                        const v3 qq = vcross(s, edge_1);
                        const float v = f * vdot(cd, qq);
                        const float t = f * vdot(edge_2, qq);
                        const bool hit = (u > 0.0f) & (u < 1.0f) & (v > 0.0f) & ((u + v) < 1.0f) & (t > RAY_EPSILON) & (t < W.ray.tcur);
                        const unsigned long long hm = __ballot(hit) & W.m;
                        if (hm != 0ull) {
                            const bool mine = (hm >> lane) & 1ull;
                            if (ANY) { if (mine) W.occluded = true; W.alive &= ~hm; W.m &= ~hm; if (W.m == 0ull) break; }
                            else if (mine) { W.ray.tcur = t; W.best_u = u; W.best_v = v; W.best_tri = i; W.best_prim = PRIM_TRI | W.cur_inst; }
                        }
                    }
                } else {
                    // TLAS leaf: the reference enters the instances first .. first+count-1 in order (TopLevelBVH.cpp:71-75);
                    // one iterator entry instead of `count` entries keeps the packet stack bounded
                    pk_push<ANY>(W, W.cur_first, leafc, W.m, 0, PK_FLAG_ITER);
                }
                W.m = 0ull;
                PK_T1(2);
            } else if (ASM && FAST) {
                PK_T0();
                pk_descend<ANY>(W, sc.stack_size, ordered ? 1 : 0, (!ANY && W.floor_sp >= 0) ? H.t0 + (W.pk_steps >> H.grow) : -1);      // inner nodes, hand-scheduled: down to a leaf (or m == 0, or too few lanes)
                PK_T1(1);
            } else {
                // inner node.  Reference stack: `depth` pending entries, two pushed before one is popped (BottomLevelBVH.cpp:381-387)
                if (W.depth + 2 > sc.stack_size) W.err = 1;
                int left_first = 1;                                      // BVHNode::should_visit_left_first, BVHNode.h:30-40 (wave-uniform int: scalar selects)
                if (ordered) {
                    const int axis = (int)((uint32_t)W.cur_cnt >> 30);
                    if (axis != 0) {
                        const float c = axis == 1 ? W.ray.d.x : (axis == 2 ? W.ray.d.y : W.ray.d.z);
                        const unsigned long long pos = __ballot(c > 0.0f) & W.m;
                        if (ANY) left_first = (int)((pos >> (__ffsll((long long)W.m) - 1)) & 1ull);      // any order is exact: the first lane's
                        else if (pos == 0ull) left_first = 0;
                        else if (pos != W.m) {
                            // lanes disagree on the order: park the right-first lanes on this node, walk the left-first lanes now
                            pk_push<ANY>(W, W.cur_first, W.cur_cnt, W.m & ~pos, W.cur_idx, W.depth | PK_FLAG_PRETESTED);
                            W.m = pos;
                        }
                    }
                }
                left_first = rfl(left_first);
                const int left = W.cur_first;
                const float4 la = sld(W.nodes, 2 * left), lb = sld(W.nodes, 2 * left + 1), ra = sld(W.nodes, 2 * left + 2), rb = sld(W.nodes, 2 * left + 3);
                if (PSTAT) { W.st_pairs++; W.st_pair_lanes += (unsigned long long)__popcll(W.m); }
                const unsigned long long ml = __ballot(pk_slab<FAST>(la, lb, W.ray)) & W.m, mr = __ballot(pk_slab<FAST>(ra, rb, W.ray)) & W.m;
                const unsigned long long m_near = left_first ? ml : mr, m_far = left_first ? mr : ml;
                const int near_idx = left_first ? left : left + 1, far_idx = left_first ? left + 1 : left;
                const int near_first = left_first ? as_i(lb.z) : as_i(rb.z), near_cnt = left_first ? as_i(lb.w) : as_i(rb.w);
                const int far_first  = left_first ? as_i(rb.z) : as_i(lb.z), far_cnt  = left_first ? as_i(rb.w) : as_i(lb.w);
                if (m_near != 0ull) {
                    if (m_far != 0ull) pk_push<ANY>(W, far_first, far_cnt, m_far, far_idx, W.depth);       // popped with `depth` entries pending
                    W.cur_idx = rfl(near_idx); W.cur_first = rfl(near_first); W.cur_cnt = rfl(near_cnt); W.m = m_near; W.depth = rfl(W.depth + 1);   // far sibling pending below us
                } else if (m_far != 0ull) {
                    W.cur_idx = rfl(far_idx); W.cur_first = rfl(far_first); W.cur_cnt = rfl(far_cnt); W.m = m_far;   // near popped and failed, far popped: depth unchanged
                } else W.m = 0ull;
            }
            continue;
        }
        // ---------------- pop ----------------
        PK_T0();
        if (ANY && W.alive == 0ull) return true;
        if (W.floor_sp >= 0 && W.sp == W.floor_sp) {            // BottomLevelBVH::trace / intersect returned: back to world space
            if (!ANY && ASM && FAST && W.pub != W.alive) {       // closest-hit hybrid: the private lanes finish this instance on their own
                if (W.lane_nodes4c) pk_lane_phase_closest<true>(W, lane, ordered, H, W.lane_nodes, W.lane_nodes4c, W.lane_tris);
                else pk_lane_phase_closest<false>(W, lane, ordered, H, W.lane_nodes, nullptr, W.lane_tris);
                W.pub = W.alive;
            }
            if (W.sp == 0) return true;                         // nothing left in the TLAS: the world-space ray is not needed again
            W.floor_sp = -1; W.nodes = W.tlas;
            v3 o, d; world(o, d);
            pk_set_ray(W.ray, o, d, vrcp(d));                   // finite for every lane of a packet that started FAST
        }
        if (W.sp == 0) return true;
        W.sp--;
        const int ea = __builtin_amdgcn_readlane(W.st_a, W.sp), eb = __builtin_amdgcn_readlane(W.st_b, W.sp);
        const int edf = __builtin_amdgcn_readlane(W.st_df, W.sp);
        const unsigned long long em = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(W.st_mhi, W.sp) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(W.st_mlo, W.sp);
        unsigned long long pm = em & W.alive;
        if (!ANY && ASM && FAST && (pm & ~W.pub) != 0ull) {
            // private lanes: the entry goes to their work lists (to be re-tested at its turn unless it is a pretested split entry)
            if (((pm & ~W.pub) >> lane) & 1ull) {
                H.fifo[W.fifo_n * H.fifo_stride] = make_int2(__builtin_amdgcn_readlane(W.st_idx, W.sp), (edf & PK_FLAG_PRETESTED) ? as_i(PK_KEY_PASSED) : 0x7fc00000); W.fifo_n++;
            }
            pm &= W.pub;
        }
        if (pm == 0ull) continue;
        if (edf & PK_FLAG_ITER) {
            if (eb > 1) pk_push<ANY>(W, ea + 1, eb - 1, pm, 0, PK_FLAG_ITER);
            // Mesh::trace / Mesh::intersect (Mesh.cpp:17-40): ray into model space
            const int inst = rfl(sc.tlas_indices[ea]);
            const rtx_instance & I = sc.instances[inst];
            v3 o, d; world(o, d);
            const v3 co = xform_pos(I.world_inv, o), cd = xform_dir(I.world_inv, d), cinv = vrcp(cd);
            pk_set_ray(W.ray, co, cd, cinv);
            W.cur_blas = rfl(I.blas_id);
            const DevBlas & B = sc.blas[W.cur_blas];
            W.nodes = pk_cptr(B.pk_nodes); W.tris = pk_cptr(B.tri_hot); W.cur_inst = inst;
            W.lane_nodes = RTX_GPTR(B.nodes); W.lane_tris = RTX_GPTR(B.tri_hot); if (!ANY) W.lane_nodes4c = RTX_GPTR(B.pk4c_nodes);
            W.floor_sp = W.sp; W.depth = 0;
            const float4 na = sld(W.nodes, 0), nb = sld(W.nodes, 1);            // BLAS root (BottomLevelBVH.cpp:360-370)
            const bool risky = FAST && __ballot(W.valid && !inv_is_finite(cinv) && pk_nan_possible(B, co, cinv)) != 0ull;
            W.m = (risky ? __ballot(pk_slab<false>(na, nb, W.ray)) : __ballot(pk_slab<FAST>(na, nb, W.ray))) & pm;
            W.cur_idx = 0; W.cur_first = as_i(nb.z); W.cur_cnt = as_i(nb.w);
            if (risky) return false;
            if (ASM && FAST && !ANY && H.closest_asm) {
                // the shared part of the closest-hit walk of this instance, hand-scheduled; returns with the packet stack back at floor_sp and
                // m == 0: the pop below runs the private lanes' phase and returns to world space
#ifdef RTX_LANE_PROF
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long sw_t0 = __builtin_amdgcn_s_memtime(); const int sw_s0 = W.pk_steps;
#endif
#ifdef RTX_PK_SKIPWALK
                W.m = 0ull;                                          // measurement only: everything but the BLAS walk
#else
                pk_blas_closest_asm(W, sc.stack_size, ordered ? 1 : 0, H, PRIM_TRI | inst, H.gthread);
#endif
#ifdef RTX_LANE_PROF
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                W.lp[13] = (uint32_t)rfl((int)(W.lp[13] + (uint32_t)(__builtin_amdgcn_s_memtime() - sw_t0))); W.lp[14] = (uint32_t)rfl((int)(W.lp[14] + (uint32_t)(W.pk_steps - sw_s0)));
                W.lp[15] = (uint32_t)rfl((int)(W.lp[15] + 1u));
#endif
            }
            if (ASM && FAST && ANY) {
                // the whole BLAS walk, hand-scheduled (returns with m == 0), then the per-lane phase over what it deferred
                PK_T0();
                W.psp = 0;
                if (SPLIT) {                                         // split walk (every mesh has 4-wide records): deferred nodes become items, no per-lane phase here
#ifdef RTX_PK_SKIPWALK
                    W.m = 0ull;                                      // measurement only: everything but the BLAS walk
#else
                    pk_blas_any_asm4(W, pk_cptr(B.pk4_nodes), H.lane_lds, H.t0, H.t_leaf, H.grow, H.item_cap, H.item_base, W.item_slot);
#endif
                    PK_T1(1);
                } else if (B.pk4_nodes) {                            // nested boxes: the 4-wide records
                    pk_blas_any_asm4(W, pk_cptr(B.pk4_nodes), H.lane_lds, H.t0, H.t_leaf, H.grow, 0u, 0ull, 0u);
                    PK_T1(1);
                    if (W.alive != 0ull && __ballot(W.psp > 0) != 0ull) { PK_T0(); pk_lane_phase_any4(W, lane, H.lane_lds, H.spill, H.spill_stride, RTX_GPTR(B.pk4_nodes), RTX_GPTR(B.tri_hot)); PK_T1(2); }
                } else {
                    pk_blas_any_asm(W, sc.stack_size, H.lane_lds, H.t0, H.grow);
                    PK_T1(1);
                    if (W.alive != 0ull && __ballot(W.psp > 0) != 0ull) { PK_T0(); pk_lane_phase_any(W, lane, H.stk, H.spill, H.spill_stride, RTX_GPTR(B.nodes), RTX_GPTR(B.tri_hot)); PK_T1(2); }
                }
            }
        } else {
            W.m = pm;
            if (!ANY && !(edf & PK_FLAG_PRETESTED)) {
                // far child: the reference tests it now, against each lane's current closest distance
                const int eidx = __builtin_amdgcn_readlane(W.st_idx, W.sp);
                const float4 na = sld(W.nodes, 2 * eidx), nb = sld(W.nodes, 2 * eidx + 1);
                W.m = __ballot(pk_slab<FAST>(na, nb, W.ray)) & pm;
                W.cur_idx = eidx;
            } else if (!ANY) W.cur_idx = __builtin_amdgcn_readlane(W.st_idx, W.sp);
            W.cur_first = rfl(ea); W.cur_cnt = rfl(eb); W.depth = rfl(edf & 0xffff);
        }
        PK_T1(3);
    }
}

// FUSE (closest-hit only): the wave shades the hits of its packet itself (shade_ray, rtx_shade.h) instead of writing hit records for a
// k_shade launch.  Why: with several frames in flight a k_shade launch (168 VGPRs per wave) waits for room beside the other frames'
// persistent traversal grids — 0.07 ms of work took 0.29 ms per launch, a third of a frame's time on its stream — while a wave that
// already owns its slot simply carries on; the hit records' round trip through HBM (20 B per ray each way) and four launches per frame go
// away too.  Child-ray slots: one atomic per packet that spawns any (wave_alloc2).
template <bool ANY, bool PSTAT, bool SPLIT = false, bool FUSE = false, bool CULL = false>
__global__ __launch_bounds__(RTX_PK_BLOCK, ANY ? (SPLIT ? RTX_PK_WAVES_SPLIT : RTX_PK_WAVES_ANY) : (FUSE ? RTX_PK_WAVES_FUSED : RTX_PK_WAVES_CLOSEST))
void k_packet(const DevScene sc, const DevQueues q, const int level, const int level_hi) {
    // hybrid any-hit walk: the lanes' private stacks, striped through LDS ([entry][lane]: conflict-free)
    // closest-hit: RTX_LDS_STACK entries + their pop-time keys (the per-lane kernels' stack layout, rtx_trace.h)
    __shared__ int s_priv[RTX_PK_BLOCK / RTX_WAVE][SPLIT ? 1 : (ANY ? RTX_PK_PRIV : 2 * RTX_LDS_STACK)][RTX_WAVE];
    const int lane = threadIdx.x & 63;
    PkHybrid H;
    H.stk = &s_priv[threadIdx.x >> 6][0][lane];
    H.stk_key = (float *)&s_priv[threadIdx.x >> 6][(ANY || SPLIT) ? 0 : RTX_LDS_STACK][lane];
    H.lane_lds = (uint32_t)(uintptr_t)H.stk;
    H.spill_stride = q.spill_threads;
    {   // spill regions of rtx_trace.h: [0, S) closest entries, [S, 2S) closest keys, [2S, 3S) shadow-ray entries
        const size_t region = (size_t)q.spill_threads * (RTX_MAX_STACK - RTX_LDS_STACK), gt = blockIdx.x * blockDim.x + threadIdx.x;
        H.spill = q.spill + (ANY ? 2 * region : 0) + gt;
        H.spill_key = (float *)(q.spill + region) + gt;
        H.fifo = (int2 *)q.pk_fifo + gt; H.fifo_stride = q.spill_threads; H.gthread = (uint32_t)gt;
    }
    H.closest_asm = q.pk_closest_asm;
    H.t0 = ANY ? q.pk_defer_t0 : (level == 0 ? q.pk_defer_t0_primary : q.pk_defer_t0_closest); H.t_leaf = q.pk_defer_leaf; H.grow = q.pk_defer_grow;
    H.item_cap = 0; H.item_base = 0ull;
    const uint32_t gwave  = (uint32_t)rfl((int)(blockIdx.x * (RTX_PK_BLOCK / RTX_WAVE) + (threadIdx.x >> 6)));
    const uint32_t nwaves = gridDim.x * (RTX_PK_BLOCK / RTX_WAVE);

    if (ANY && SPLIT && q.pk_items) { H.item_cap = (uint32_t)q.pk_item_cap; H.item_base = (unsigned long long)(q.pk_items + 3 * (size_t)gwave * (size_t)q.pk_item_cap); }
    DevCounters * const ctr = q.counters;
    const int nL = sc.light_count;
    // packets: closest-hit = consecutive 64-slot groups of the level's queue; any-hit = per (level, light) segment, each padded
    // to whole packets so that a packet never mixes lights or levels
    uint32_t npk = 0;
    if (ANY) { for (int d = level; d <= level_hi; d++) npk += ((shadow_slots(ctr, q, d) + 63u) >> 6) * (uint32_t)nL; }
    else npk = (ctr->ray_count[level] + 63u) >> 6;
    npk = (uint32_t)rfl((int)npk);
    uint32_t * const heads = q.pk_heads + (size_t)((ANY ? RTX_MAX_LEVELS + 1 : 0) + level) * RTX_PK_CLASSES * 32;
    const bool ordered = sc.traversal_strategy == RTX_TRAVERSE_ORDERED;

    PkWalk W;
#ifdef RTX_PK_PROF
    for (int i = 0; i < 10; i++) W.prof[i] = 0ull;
    const unsigned long long pk_k0_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef RTX_LANE_PROF
    for (int i = 0; i < 16; i++) W.lp[i] = 0u;
    const unsigned long long lp_k0 = __builtin_amdgcn_s_memtime(), lp_r0 = __builtin_amdgcn_s_memrealtime(); bool lp_had_packet = false;
#endif
    W.tlas = pk_cptr(sc.pk_tlas_nodes);
    W.err = 0; W.st_pairs = W.st_tris = W.st_pair_lanes = W.st_tri_lanes = 0ull; W.item_n = 0; W.item_slot = 0; W.cur_blas = 0;
    W.st_a = W.st_b = W.st_mlo = W.st_mhi = W.st_idx = W.st_df = 0;
    unsigned long long st_packets = 0ull;
    ShadeTally tally = { 0, 0, 0, 0, 0, 0, 0, 0 };               // FUSE: this wave's ray tallies over all its packets
    // Child-ray slots come out of wave-private CHUNKS of the next level's queue: one atomic per RTX_PK_CHUNK slots instead of one per packet
    // (the queue counter sustains ~90 atomics/us: 32 400 packets of a 1080p level-0 launch, each waiting for its own atomic, cost +0.24 ms).
    // A packet's children fill what is left of the wave's chunk and continue in a fresh one; the slots a wave has not used when it exits
    // are marked as holding no ray (RAY_DEAD_PIXEL), which every consumer treats like a clipped pixel.
    uint32_t chunk_next = 0, chunk_end = 0;
    auto wave_alloc2 = [&](bool want_a, bool want_b, uint32_t * counter, uint32_t & slot_a, uint32_t & slot_b) {
        const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
        const uint32_t ca = (uint32_t)__popcll(ma), need = ca + (uint32_t)__popcll(mb);
        if (need == 0u) { slot_a = slot_b = 0u; return; }
        const uint32_t avail = chunk_end - chunk_next, first_base = chunk_next, first_n = need < avail ? need : avail;
        uint32_t second_base = 0;
        if (need > avail) {
            const uint32_t want = need - avail, grab = (want + RTX_PK_CHUNK - 1u) / RTX_PK_CHUNK * RTX_PK_CHUNK;
            if (lane == 0) second_base = atomicAdd(counter, grab);
            second_base = (uint32_t)rfl((int)second_base);
            chunk_next = second_base + want; chunk_end = second_base + grab;
        } else chunk_next += need;
        const uint32_t rank_a = (uint32_t)__popcll(ma & ((1ull << lane) - 1ull)), rank_b = ca + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
        slot_a = rank_a < first_n ? first_base + rank_a : second_base + (rank_a - first_n);
        slot_b = rank_b < first_n ? first_base + rank_b : second_base + (rank_b - first_n);
    };
    uint32_t cls = gwave & (RTX_PK_CLASSES - 1);
    int probes_left = 2;                       // own head, then one neighbour (steal), then stop
    bool first_round = true;
    const bool all_static = nwaves >= npk;     // every packet is covered by the static first round: no atomics at all

    for (;;) {
        // ---------------- next packet ----------------
        PK_T0();
        uint32_t pk;
        if (first_round) { first_round = false; pk = gwave; if (pk >= npk) break; }      // static first round: wave w takes packet w
        else {
            if (all_static) break;
            bool got = false;
            while (probes_left > 0) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&heads[cls * 32], 1u);
                k = (uint32_t)rfl((int)k);
                pk = k * RTX_PK_CLASSES + cls;
                if (pk < npk) { got = true; break; }
                cls = (cls + 1) & (RTX_PK_CLASSES - 1); probes_left--;
            }
            if (!got) break;
        }

#ifdef RTX_LANE_PROF
        lp_had_packet = true;
#endif
        // level-0 closest-hit packets: longest first, by what the same packet cost in the previous frame of this context (scheduling only: a wave that
        // draws a 600 us packet last sets the launch's tail)
        const unsigned long long pk_t_start = (!ANY && level == 0 && q.pk_cost) ? __builtin_amdgcn_s_memtime() : 0ull;
        if (!ANY && level == 0 && q.pk_lpt_order) pk = (uint32_t)rfl((int)q.pk_lpt_order[pk]);
        // shadow-ray launches walk their packets last level first: the packets of the deeper levels are the incoherent, expensive ones
        // (longest jobs first: 0.79 -> 0.73 ms for the merged launch); RTX_PK_ORDER=0 restores queue order
        if (ANY && q.pk_order != 0) pk = npk - 1u - pk;
        // ---------------- the packet's rays ----------------
        W.valid = false; W.occluded = false;
        uint32_t out_slot = 0;
        v3 wo = V3(0.0f, 0.0f, 0.0f), wd = V3(1.0f, 1.0f, 1.0f);
        W.ray.tcur = INFINITY;
        W.best_prim = PRIM_NONE; W.best_tri = 0; W.best_u = 0.0f; W.best_v = 0.0f;
        bool clipped = false;
        uint32_t pix_idx = 0;
        int any_level = 0, any_light = 0; uint32_t any_rank = 0;
        if (ANY) {
            uint32_t local = pk; int d = level;
            uint32_t lit_n = shadow_slots(ctr, q, d), seg = (lit_n + 63u) >> 6;
            while (d < level_hi && local >= seg * (uint32_t)nL) { local -= seg * (uint32_t)nL; d++; lit_n = shadow_slots(ctr, q, d); seg = (lit_n + 63u) >> 6; }
            uint32_t light = 0; while (local >= seg && light + 1u < (uint32_t)nL) { local -= seg; light++; }      // wave-uniform, at most light_count - 1 rounds: no integer division on the VALU
            const uint32_t rank = local * 64u + (uint32_t)lane;
            W.valid = rank < lit_n && shadow_slot_is_lit(q, d, rank);      // the shadow rays of the 64 rays of the level's packet `local % seg`
            any_level = d; any_light = (int)light; any_rank = rank;
            if (W.valid) {
                out_slot = (uint32_t)q.shadow_base[d] + light * (uint32_t)q.level_cap[d] + rank;
                load_shadow_ray(sc, q, d, (int)light, rank, out_slot, wo, wd, W.ray.tcur);      // derived from the hit's point and the packet's light (wave-uniform: scalar loads)
                if (W.ray.tcur < 0.0f) W.valid = false;          // culled by k_shade<.., CULL>: contribution exactly zero, already resolved
            }
        } else {
            pix_idx = pk * 64u + (uint32_t)lane;
            W.valid = pix_idx < ctr->ray_count[level];
            if (W.valid) {
                out_slot = (uint32_t)q.level_base[level] + pix_idx;
                if (level == 0) {
                    int px, py;
                    if (primary_pixel(sc, q, pix_idx, px, py)) { RayFull r; primary_ray(sc, px, py, r, false); wo = r.o; wd = r.d; }
                    else { clipped = true; W.valid = false; }
                } else {
                    const float4 r0 = q.r0[out_slot], r1 = q.r1[out_slot];
                    if (as_i(r1.z) == RAY_DEAD_PIXEL) { clipped = true; W.valid = false; }      // unused slot of a wave's chunk (below): no ray
                    else { wo = V3(r0.x, r0.y, r0.z); wd = V3(r0.w, r1.x, r1.y); }
                }
            }
        }
        // the world-space ray is not kept in registers while an instance is walked: it is reloaded (or regenerated) when the walk
        // returns to the TLAS and when the next instance is entered
        auto world = [&](v3 & o, v3 & d) {
            o = V3(0.0f, 0.0f, 0.0f); d = V3(1.0f, 1.0f, 1.0f);
            if (!W.valid) return;
            if (ANY) { float tm; load_shadow_ray(sc, q, any_level, any_light, any_rank, out_slot, o, d, tm); }
            else if (level == 0) { int px, py; primary_pixel(sc, q, pix_idx, px, py); RayFull r; primary_ray(sc, px, py, r, false); o = r.o; d = r.d; }
            else { const float4 r0 = q.r0[out_slot], r1 = q.r1[out_slot]; o = V3(r0.x, r0.y, r0.z); d = V3(r0.w, r1.x, r1.y); }
        };
        // Scene::trace_primitives / intersect_primitives: spheres, then planes, then the TLAS (Scene.cpp:173-190)
        if (W.valid) {
            for (int i = 0; i < sc.sphere_count; i++) {
                if (ANY) { if (sphere_any(sc.spheres[i], wo, wd, W.ray.tcur)) { W.occluded = true; break; } }
                else { float t; if (sphere_closest(sc.spheres[i], wo, wd, W.ray.tcur, t)) { W.ray.tcur = t; W.best_prim = PRIM_SPHERE | i; } }
            }
            if (!W.occluded) for (int i = 0; i < sc.plane_count; i++) {
                const float t = plane_t(sc.planes[i], wo, wd);
                if ((t > RAY_EPSILON) & (t < W.ray.tcur)) { if (ANY) { W.occluded = true; break; } else { W.ray.tcur = t; W.best_prim = PRIM_PLANE | i; } }
            }
        }
        W.psp = 0; W.pk_steps = 0; W.fifo_n = 0; W.item_slot = out_slot;
        W.alive = __ballot(W.valid && !W.occluded);     // any-hit: lanes leave when occluded; closest-hit: constant
        const unsigned long long alive0 = W.alive;
        W.pub = W.alive;
        if (PSTAT) st_packets++;

        if (W.alive != 0ull && sc.tlas_node_count > 0) {
            const v3 winv = vrcp(wd);
            pk_set_ray(W.ray, wo, wd, winv);
            const bool nanfree_world = __ballot(W.valid && !inv_is_finite(winv) && pk_nan_possible_tlas(sc, wo, winv)) == 0ull;
            W.nodes = W.tlas; W.tris = W.tlas;
            W.sp = 0; W.floor_sp = -1; W.cur_inst = 0; W.depth = 0; W.cur_idx = 0;
            {   // TLAS root: pushed, popped and tested by the reference (TopLevelBVH.cpp:58-67)
                const float4 na = sld(W.nodes, 0), nb = sld(W.nodes, 1);
                W.m = (nanfree_world ? __ballot(pk_slab<true>(na, nb, W.ray)) : __ballot(pk_slab<false>(na, nb, W.ray))) & W.alive;
                W.cur_first = as_i(nb.z); W.cur_cnt = as_i(nb.w);
            }
            PK_T1(0);
#ifdef RTX_PK_PROF
            const unsigned long long pk_w0_ = __builtin_amdgcn_s_memtime();
#endif
            bool done = false;
            if (nanfree_world && !done) done = pk_walk<ANY, PSTAT, true, !PSTAT, SPLIT>(sc, W, lane, ordered, H, world);
            if (!done) pk_walk<ANY, PSTAT, false, false, SPLIT>(sc, W, lane, ordered, H, world);
#ifdef RTX_PK_PROF
            {   // slot 4: the longest single packet walk (max), slot 9: walks longer than 2^18 ticks
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - pk_w0_;
                if (dt > W.prof[4]) W.prof[4] = dt;
            }
#endif
        }

        // ---------------- results ----------------
        if (ANY) { if (((alive0 & ~W.alive) >> lane) & 1ull) W.occluded = true; }      // lanes that left the packet were hit
        if (ANY) { if (W.valid) q.socc[out_slot] = W.occluded ? 1u : 0u; }
        else if (FUSE) {
            // Raytracer::bounce from the hit on (Raytracer.cpp:104-346), for the packet's 64 rays at once; a clipped pixel's slot exists but holds no ray
            const bool in_range = W.valid || clipped;
#ifdef RTX_LANE_PROF
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long sp_t0 = __builtin_amdgcn_s_memtime();
#endif
            shade_ray<false, CULL>(sc, q, level, pix_idx, in_range, W.valid, make_float4(W.ray.tcur, W.best_u, W.best_v, as_f(W.best_prim)),
                                   clipped ? -1 : W.best_tri, wave_alloc2, tally);
#ifdef RTX_LANE_PROF
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            W.lp[13] = (uint32_t)rfl((int)(W.lp[13] + (uint32_t)(__builtin_amdgcn_s_memtime() - sp_t0))); W.lp[14] = (uint32_t)rfl((int)(W.lp[14] + 1u));
#endif
        }
        else if (W.valid) { q.h0[out_slot] = make_float4(W.ray.tcur, W.best_u, W.best_v, as_f(W.best_prim)); q.h1[out_slot] = W.best_tri; }
        else if (clipped) { q.h0[out_slot] = make_float4(INFINITY, 0.0f, 0.0f, as_f(PRIM_NONE)); q.h1[out_slot] = -1; }   // clipped pixel: never shaded
        if (!ANY && level == 0 && q.pk_cost && lane == 0) q.pk_cost[pk] = (uint32_t)((__builtin_amdgcn_s_memtime() - pk_t_start) >> 6);
    }
    if (!ANY && FUSE && chunk_next < chunk_end) {      // the unused tail of this wave's last chunk (< RTX_PK_CHUNK slots): no ray there
        const uint32_t dead = chunk_next + (uint32_t)lane;
        if (dead < chunk_end && dead < (uint32_t)q.level_cap[level + 1])
            q.r1[(uint32_t)q.level_base[level + 1] + dead] = make_float4(0.0f, 0.0f, as_f(RAY_DEAD_PIXEL), as_f(-1));
    }
    if (!ANY && FUSE && lane < 4) {        // this wave's share of the level's ray tallies (every wave writes its entry: q.stats_n[level] = waves of this launch)
        const uint32_t v = lane == 0 ? tally.n_primary : (lane == 1 ? tally.n_shadow : (lane == 2 ? tally.n_refl : tally.n_refr));
        q.stats_partial[((size_t)level * q.stats_stride + gwave) * 4 + lane] = v;
    }

#ifdef RTX_PK_PROF
    W.prof[6] = __builtin_amdgcn_s_memtime() - pk_k0_;          // this wave's lifetime
    W.prof[7] = __builtin_amdgcn_s_memtime();                    // last end (max)
    W.prof[8] = (1ull << 62) - pk_k0_;                           // first start (max of the complement)
    if (lane == 0) for (int i = 0; i < 10; i++) { if (i == 4 || i == 5 || i == 7 || i == 8 || i == 9) atomicMax(&ctr->work[(ANY ? 0 : 10) + i], W.prof[i]); else atomicAdd(&ctr->work[(ANY ? 0 : 10) + i], W.prof[i]); }
#endif
#ifdef RTX_LANE_PROF
    if (!ANY && level == q.prof_level % 100 && lane == 0 && (W.lp[12] > 0u || (q.prof_level >= 100 && lp_had_packet))) {       // waves that walked something (no atomic storm of the idle ones); + 100: every wave that had a packet
        // wave-uniform values; inner-wait ticks were accumulated by every lane identically
        for (int i = 0; i < 10; i++) atomicAdd(&ctr->work[i], (unsigned long long)W.lp[i]);
        atomicMax(&ctr->work[10], (unsigned long long)W.lp[10]); atomicMax(&ctr->work[11], (unsigned long long)W.lp[11]); atomicAdd(&ctr->work[12], (unsigned long long)W.lp[12]);
        atomicMax(&ctr->work[13], __builtin_amdgcn_s_memtime() - lp_k0);      // longest wave lifetime
        atomicAdd(&ctr->work[14], __builtin_amdgcn_s_memtime() - lp_k0);
        atomicAdd(&ctr->work[15], 1ull);                                        // waves
        atomicAdd(&ctr->work[16], __builtin_amdgcn_s_memrealtime() - lp_r0);
        atomicAdd(&ctr->work[17], (unsigned long long)W.lp[13]); atomicAdd(&ctr->work[18], (unsigned long long)W.lp[14]);      // shared (asm) walks: ticks, node visits (FUSE builds: fused shading ticks, packets)
        atomicAdd(&ctr->work[19], (unsigned long long)W.lp[15]);                                                                // shared walks    // the same lifetimes on the constant 100 MHz clock: calibrates the s_memtime tick
    }
#endif
    if (ANY && SPLIT && q.pk_items && lane == 0) { q.pk_item_count[gwave] = (uint32_t)W.item_n; if ((uint32_t)W.item_n > __hip_atomic_load(&ctr->item_max[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctr->item_max[level], (uint32_t)W.item_n); }      // few waves raise the maximum: one counter sustains only ~90 atomics/us
    if (W.err) { if (lane == 0) atomicOr(&ctr->error_flags, ERR_STACK_OVERFLOW); }
    if (PSTAT && lane == 0) {
        atomicAdd(&ctr->work[15 + PKS_PACKETS], st_packets); atomicAdd(&ctr->work[15 + PKS_PAIR_STEPS], W.st_pairs);
        atomicAdd(&ctr->work[15 + PKS_TRI_STEPS], W.st_tris); atomicAdd(&ctr->work[15 + PKS_PAIR_LANES], W.st_pair_lanes);
        atomicAdd(&ctr->work[15 + PKS_TRI_LANES], W.st_tri_lanes);
    }
}

// ---- split walk, second half: the items the shadow-ray packet kernel emitted, one item per lane ----------------------------------------
// An item = (model-space origin, max distance) (model-space direction, ray slot) (node that passed the ray's test, mesh): 48 bytes, written by
// the packet walk where few lanes wanted a node.  Most are tiny (2.5 loop iterations on average in the cfg3 frame, a few take a hundred) and
// there are millions of them.  Unit of work = 64 consecutive items of one chunk (chunk c = what wave c of the packet launch emitted); units are
// numbered batch-major (unit u = batch u / chunks of chunk u % chunks, as many batches as the fullest chunk has) and dealt round-robin to the waves of a persistent grid, so every
// wave gets the same mix of full, partial and empty units without any queue.  Inside a wave, a lane that finishes its item takes the next
// one of the wave's units at once (ballot + prefix rank): a long item keeps one lane busy, not 64.  A lane walks its item's subtree over the
// 4-wide records with a private stack (LDS, then scratch); a hit sets the ray's occlusion flag (the packet kernel wrote 0 or 1 before); a
// ray already flagged skips its items.
#define RTX_ITEM_LDS 8
#ifndef RTX_ITEM_BLOCK
#define RTX_ITEM_BLOCK 64
#endif
#define RTX_ITEM_MESHES 32
#ifndef RTX_ITEM_WAVES
#define RTX_ITEM_WAVES 6
#endif
__global__ __launch_bounds__(RTX_ITEM_BLOCK, RTX_ITEM_WAVES)
void k_items(const DevScene sc, const DevQueues q, const int n_chunks, const int level) {
    __shared__ int s_stk[RTX_ITEM_LDS][RTX_ITEM_BLOCK];
    __shared__ unsigned long long s_mesh[RTX_ITEM_MESHES][2];          // (4-wide records, triangles) of the first meshes: a hand-over costs no dependent global fetch for them
    for (int b = threadIdx.x; b < RTX_ITEM_MESHES && b < sc.blas_count; b += blockDim.x) { s_mesh[b][0] = (unsigned long long)sc.blas[b].pk4_nodes; s_mesh[b][1] = (unsigned long long)sc.blas[b].tri_hot; }
    __syncthreads();
    const uint32_t gwave = (uint32_t)rfl((int)(blockIdx.x * (RTX_ITEM_BLOCK / RTX_WAVE) + (threadIdx.x >> 6)));
    const uint32_t n_batches = ((uint32_t)rfl((int)q.counters->item_max[level]) + 63u) >> 6;      // the fullest chunk of the packet launch bounds the units
    const uint32_t nwaves = gridDim.x * (RTX_ITEM_BLOCK / RTX_WAVE), units = (uint32_t)n_chunks * n_batches;
    int ovf[RTX_PK4_MAX_NEED + 4];                                                   // stack entries beyond the LDS ones (scratch; rare)
#ifdef RTX_ITEM_STATS
    unsigned long long st_iters = 0, st_busy = 0, st_batches = 0, st_items = 0;
#endif
    // the wave's stream of items: units u, u + nwaves, ...; [pos, lim) = what is left of the current unit (wave-uniform)
    uint32_t u = gwave, pos = 0, lim = 0; const uint4 * items = q.pk_items;
    auto next_unit = [&]() {
        for (; u < units; u += nwaves) {
            const uint32_t chunk = u % (uint32_t)n_chunks, first = (u / (uint32_t)n_chunks) * 64u;
            const uint32_t n = (uint32_t)rfl((int)q.pk_item_count[chunk]);
            if (first < n) { pos = first; lim = first + 64u < n ? first + 64u : n; items = q.pk_items + 3 * (size_t)chunk * (size_t)q.pk_item_cap; u += nwaves; return true; }
        }
        return false;
    };
    bool have_unit = next_unit();
    // per lane: the item being walked
    bool busy = false, have_cur = false; uint32_t slot = 0; int psp = 0, cur_first = 0, cur_cnt = 0;
    v3 co = V3(0.0f, 0.0f, 0.0f), cd = V3(1.0f, 1.0f, 1.0f);
    PkRay ray; pk_set_ray(ray, co, cd, cd); ray.tcur = 0.0f;
    rtx_gptr nodes4 = nullptr, tris = nullptr;
    auto push = [&](int e) { if (psp < RTX_ITEM_LDS) s_stk[psp][threadIdx.x] = e; else if (psp < RTX_ITEM_LDS + RTX_PK4_MAX_NEED + 4) ovf[psp - RTX_ITEM_LDS] = e; psp++; };
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        if (idle != 0ull && have_unit) {
            const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t take = (uint32_t)__popcll(idle), avail = lim - pos;
            if (!busy && rank < avail) {
                const uint4 * const it = items + 3 * (size_t)(pos + rank);
                const uint4 q0 = it[0], q1 = it[1]; const uint2 q2 = *(const uint2 *)(it + 2);
                slot = q1.w;
                if (q2.y < RTX_ITEM_MESHES) { nodes4 = (rtx_gptr)s_mesh[q2.y][0]; tris = (rtx_gptr)s_mesh[q2.y][1]; }
                else { const DevBlas & B = sc.blas[q2.y]; nodes4 = RTX_GPTR(B.pk4_nodes); tris = RTX_GPTR(B.tri_hot); }
                co = V3(as_f((int)q0.x), as_f((int)q0.y), as_f((int)q0.z)); cd = V3(as_f((int)q1.x), as_f((int)q1.y), as_f((int)q1.z));
                pk_set_ray(ray, co, cd, vrcp(cd)); ray.tcur = as_f((int)q0.w);        // vrcp as at the instance entry of the packet walk: the same bits
                cur_first = (int)(q2.x & 0x00ffffffu); cur_cnt = (int)(q2.x >> 24); have_cur = true; psp = 0;
                busy = q.socc[slot] == 0u;                                            // already occluded: the item is void
#ifdef RTX_ITEM_STATS
                st_items++;
#endif
            }
            pos += take < avail ? take : avail;
            if (pos >= lim) have_unit = next_unit();
        }
        if (__ballot(busy) == 0ull) { if (!have_unit) break; else continue; }
#ifdef RTX_ITEM_STATS
        st_iters++; st_busy += (unsigned long long)__popcll(__ballot(busy));
#endif
        if (busy && have_cur && cur_cnt == 0) {
            const int base = 4 * cur_first;
            const float4 a0 = gld(nodes4, base), a1 = gld(nodes4, base + 1), b0 = gld(nodes4, base + 2), b1 = gld(nodes4, base + 3);
            const float4 c0 = gld(nodes4, base + 4), c1 = gld(nodes4, base + 5), d0 = gld(nodes4, base + 6), d1 = gld(nodes4, base + 7);
            const bool p0 = pk_slab<true>(a0, a1, ray), p1 = pk_slab<true>(b0, b1, ray), p2 = pk_slab<true>(c0, c1, ray), p3 = pk_slab<true>(d0, d1, ray);
            if (p3 & (p0 | p1 | p2)) push((int)(((uint32_t)as_i(d1.w) << 24) | (uint32_t)as_i(d1.z)));
            if (p2 & (p0 | p1))      push((int)(((uint32_t)as_i(c1.w) << 24) | (uint32_t)as_i(c1.z)));
            if (p1 & p0)             push((int)(((uint32_t)as_i(b1.w) << 24) | (uint32_t)as_i(b1.z)));
            have_cur = p0 | p1 | p2 | p3;
            cur_first = p0 ? as_i(a1.z) : (p1 ? as_i(b1.z) : (p2 ? as_i(c1.z) : as_i(d1.z)));
            cur_cnt   = p0 ? as_i(a1.w) : (p1 ? as_i(b1.w) : (p2 ? as_i(c1.w) : as_i(d1.w)));
        }
        if (busy && have_cur && cur_cnt > 0) {
            have_cur = false;
            for (int t = cur_first; t < cur_first + cur_cnt; t++) {
                const float4 p0 = gld(tris, RTX_TRI_STRIDE * t), e1 = gld(tris, RTX_TRI_STRIDE * t + 1), e2 = gld(tris, RTX_TRI_STRIDE * t + 2);
                const v3 edge_1 = V3(e1.x, e1.y, e1.z), edge_2 = V3(e2.x, e2.y, e2.z);
                const v3 h = vcross(cd, edge_2);
                const float a = vdot(edge_1, h);
                const float f = 1.0f / a;
                const v3 sv = vsub(co, V3(p0.x, p0.y, p0.z));
                const float uu = f * vdot(sv, h);
                if (!((uu > 0.0f) & (uu < 1.0f))) continue;
                const v3 qq = vcross(sv, edge_1);
                const float v = f * vdot(cd, qq);
                if (!((v > 0.0f) & ((uu + v) < 1.0f))) continue;
                const float tt = f * vdot(edge_2, qq);
                if (!((tt > RAY_EPSILON) & (tt < ray.tcur))) continue;
                q.socc[slot] = 1u; busy = false; break;                      // occluded
            }
        }
        if (busy && !have_cur) {
            if (psp == 0) busy = false;
            else {
                psp--;
                const int e = psp < RTX_ITEM_LDS ? s_stk[psp][threadIdx.x] : ovf[psp - RTX_ITEM_LDS];
                cur_first = e & 0x00ffffff; cur_cnt = (int)((uint32_t)e >> 24); have_cur = true;
            }
        }
    }
#ifdef RTX_ITEM_STATS
    if ((threadIdx.x & 63) == 0) { atomicAdd(&q.counters->work[16], st_iters); atomicAdd(&q.counters->work[17], st_busy); atomicMax(&q.counters->work[19], st_iters); }
    atomicAdd(&q.counters->work[15], st_items);
#endif
}
