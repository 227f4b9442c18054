// rtx_device.h — data layout in HBM shared by the kernels and the host side of librtx_hip.so.
//
// Layout rules (MI355X): everything a wave streams is "SoA at float4 granularity": lane i reads
// element i of an array of float4, so one load instruction moves 64 x 16 B = 1 KiB contiguous.
// Gathers (BVH nodes, triangles) are 16-byte aligned float4 records: a 32-byte node is two
// dwordx4 loads, a padded 48-byte triangle three.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rtx.h"

// Per-lane pointers to BVH nodes / triangles / texels come out of device tables, so the compiler cannot prove their address space and
// would emit FLAT loads (LDS-aperture check, lgkmcnt + vmcnt).  They always point into HBM: say so (global_load_dwordx4).
typedef float rtx_f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) rtx_f4v * rtx_gptr;
#define RTX_GPTR(p) ((rtx_gptr)(const void *)(p))
static __device__ __forceinline__ float4 gld(rtx_gptr p, int i) { const rtx_f4v v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
// the first 12 bytes of 16-byte record i (global_load_dwordx3: a float3 out of a padded record costs three registers, not four)
typedef float rtx_f3v __attribute__((ext_vector_type(3)));
static __device__ __forceinline__ float3 gld3(rtx_gptr p, int i) {
    const rtx_f3v v = *(const __attribute__((address_space(1))) rtx_f3v *)(p + i);
    return make_float3(v.x, v.y, v.z);
}

#define RTX_MAX_LEVELS   12          // NUMBER_OF_BOUNCES + 1 wavefront levels supported
#define RTX_TRI_STRIDE   4           // float4 per hot triangle on the device: p0, e1, e2 + padding = one 64-byte line per triangle
#ifndef RTX_LDS_STACK
#define RTX_LDS_STACK    16          // traversal stack entries kept in LDS per lane (4 KiB per wave); deeper entries spill to HBM
#endif
#define RTX_WAVE         64
#define RTX_TRACE_BLOCK  256         // 4 waves per workgroup

// prim codes stored in a hit record (w of h0)
#define PRIM_NONE     0
#define PRIM_TRI      (1 << 28)      // | instance index
#define PRIM_SPHERE   (2 << 28)      // | sphere index
#define PRIM_PLANE    (3 << 28)      // | plane index
#define PRIM_KIND(p)  ((p) & (3 << 28))
#define PRIM_INDEX(p) ((p) & ((1 << 28) - 1))

// node-record flags
#define NF_HIT      1
#define NF_LIT      2
#define NF_REFL     4
#define NF_REFR     8      // refraction_mask (|Kt|^2 > 0 and bounces left)
#define NF_TIR      16
#define NF_ENTER    32
#define RAY_DEAD_PIXEL (-1)   // ray record r1.z (pixel) of a queue slot that holds no ray: the closest-hit kernels skip it, shading marks its node NF_CLIP
#define NF_CLIP     64     // the slot holds no ray — a pixel of an edge tile outside the frame (WorkerThread.cpp:60-61) or an unused slot of a wave's chunk: nothing to resolve

struct DevBlas {
    const float4 *            nodes;      // 2 x float4 per node: (min.xyz, left_or_first) (max.xyz, count)   [per-lane kernels, rtx_trace.h]
    const float4 *            pk_nodes;   // 2 x float4 per node: (min.x, min.y, max.x, max.y) (min.z, max.z, left_or_first, count)   [packet kernels, rtx_packet.h]
    const float4 *            tri_hot;    // RTX_TRI_STRIDE x float4 per triangle: (p0, -) (e1, -) (e2, -) (padding)
    const rtx_triangle_cold * tri_cold;   // reference layout, 64 B
    const float4 *            pk4_nodes;  // 4-wide records for the shadow-ray packet walk (128 B at 64 * left-child index), or null: tree not nested / too big
    const float4 *            pk4c_nodes; // ordered 4-wide records for the closest-hit per-lane walk (build_nodes_pk4c, rtx_api.hip), or null
    const float *             planes[3];  // per axis: the distinct box-plane coordinates of the tree's nodes, ascending (plane_member, rtx_packet.h)
    int32_t node_count, tri_count, material_offset, pk4_need, pk4c_need, plane_count[3];
};

struct DevTexture {
    rtx_texture_desc desc;
    const float4 *   texels;              // ONE float4 per texel (rgb, 0): a texel is a single 16-byte load (rtx_texture.h)
};

struct DevCounters {
    uint32_t ray_count[RTX_MAX_LEVELS + 1];       // rays queued per level (level 0 = primary slots)
    uint32_t item_max[RTX_MAX_LEVELS + 1];        // split shadow-ray walk: the fullest item chunk of the packet launch that started at this level (k_items reads it)
    uint32_t fetch_closest[RTX_MAX_LEVELS + 1];   // dynamic-fetch heads of the persistent trace kernels
    uint32_t fetch_any[RTX_MAX_LEVELS + 1];
    uint32_t error_flags;                         // bit0: traversal stack overflow, bit1: queue overflow
    uint32_t pad[3];
    unsigned long long stats[4];                  // primary, shadow, reflection, refraction
    unsigned long long work[20];                  // rtx_work_counters order
};

#define ERR_STACK_OVERFLOW 1u
#define ERR_QUEUE_OVERFLOW 2u

// work[] indices (order of rtx_work_counters)
enum { W_CLOSEST_RAYS, W_ANY_RAYS, W_TLAS_C, W_TLAS_A, W_BLAS_C, W_BLAS_A, W_INST_C, W_INST_A, W_TRI_C, W_TRI_A,
       W_TRI_HITS, W_SHADED, W_SKY, W_TEXELS, W_SPAWNED };

struct DevScene {
    int32_t width, height, bounces, stack_size;
    int32_t traversal_strategy, texture_mode, mip_filter, diff_enabled;
    int32_t heatmap;                               // BVH_VISUALIZE_HEATMAP, Config.h:23
    float   max_anisotropy;
    int32_t tile_count_x;
    // camera (Camera.h:10-23)
    float cam_pos[3], cam_tl[3], cam_x[3], cam_y[3];
    float ambient[3];

    const DevBlas *       blas;       int32_t blas_count;
    const rtx_material *  materials;  int32_t material_count;
    const DevTexture *    textures;   int32_t texture_count;
    const float *         sky;        int32_t sky_size;
    const float *         ewa_table;

    const float4 *        tlas_nodes;   int32_t tlas_node_count;
    const float4 *        pk_tlas_nodes;                       // the TLAS in the packet kernels' node layout
    const int32_t *       tlas_indices; int32_t tlas_index_count;
    const rtx_instance *  instances;    int32_t instance_count;
    const rtx_sphere *    spheres;      int32_t sphere_count;
    const rtx_plane *     planes;       int32_t plane_count;
    const rtx_point_light *       point_lights; int32_t point_light_count;
    const rtx_spot_light *        spot_lights;  int32_t spot_light_count;
    const rtx_directional_light * dir_lights;   int32_t dir_light_count;
    int32_t light_count;                // total lights = shadow rays per lit hit
};

// One render call (a batch of tiles) works on P = tile_count * 1024 primary slots.  Slots of level d
// live at [level_base[d], level_base[d] + level_cap[d]) of every per-ray array below.
struct DevQueues {
    int32_t level_base[RTX_MAX_LEVELS + 1];
    int32_t level_cap[RTX_MAX_LEVELS + 1];
    int32_t shadow_base[RTX_MAX_LEVELS + 1];     // slot = shadow_base[d] + light * level_cap[d] + lit_rank
    int32_t first_tile, tile_stride, tile_count, primary_slots;

    // ray records, levels >= 1 (level 0 is regenerated from the pixel index): 5 x float4
    float4 * r0;   // o.x o.y o.z d.x
    float4 * r1;   // d.y d.z bits(pixel; RAY_DEAD_PIXEL = the slot holds no ray) bits(parent slot | refraction<<31)
    float4 * r2;   // dOdx.z dOdy.x dOdy.y dOdy.z
    float4 * r3;   // dDdx.x dDdx.y dDdx.z dDdy.x
    float4 * r4;   // dDdy.y dDdy.z dOdx.x dOdx.y
    // hit records
    float4 *  h0;  // t u v bits(prim)
    int32_t * h1;  // triangle index
    // node records (written by shade, folded by resolve)
    float4 * n0;   // albedo.xyz (or sky colour on a miss), bits(flags)
    float4 * n1;   // Ks.xyz, F_r
    float4 * n2;   // (Kt - 1).xyz, hit distance
    float4 * n3;   // unused since round 3 (pixel: from the slot at level 0; parent: ray record r1.w)
    float4 * c0;   // reflection child's returned colour (written by the child's k_resolve)
    float4 * c1;   // refraction child's returned colour, refraction child's hit distance
    // shadow rays: the hit's point and normal once per lit hit (level slot); a ray towards light L is DERIVED from the point and the light by its
    // consumers (shadow_ray_of_light, rtx_trace.h — the arithmetic of Raytracer.cpp:155-198, the same bits wherever it is evaluated), the light's
    // contribution by k_resolve for the lights that turn out unoccluded.  Per (light, hit) only the occlusion flag is stored.
    float4 *   sp; // hit point.xyz, -
    float4 *   sn; // hit normal.xyz, -
    uint32_t * socc; // 1 = occluded (with RTX_RENDER_CULL_DEAD_SHADOW_RAYS also pre-set by the shading pass: 1 = contributes nothing, do not trace; 0 = trace)
    // explicit shadow rays (rtx_debug_occluded only): origin.xyz + max distance, direction.xyz
    float4 *   s0;
    float4 *   s1;
    int32_t    shadow_explicit;   // 1: the shadow-ray kernels read s0 / s1 instead of deriving the rays
    int32_t    cull;              // 1: socc was pre-set by the shading pass (see above)

    int32_t *     spill;          // traversal stack spill, 3 regions of [(entry - RTX_LDS_STACK) * spill_threads + thread]: closest entries, closest keys, shadow entries
    int32_t       spill_threads;
    int32_t *     pk_fifo;        // closest-hit hybrid walk: per-lane work lists of (reference, key) pairs, [RTX_PK_FIFO entries][spill_threads] (rtx_packet.h)
    int32_t       prof_level;     // RTX_LANE_PROF builds: the level whose closest-hit launch leaves its phase profile in the work counters
    int32_t       pk_order;       // shadow-ray packets: 1 = last level first (default), 0 = queue order
    int32_t       pk_defer_grow;  // ... threshold = pk_defer_t0 + (steps of this packet walk >> pk_defer_grow)
    int32_t       pk_closest_asm; // 1: the shared part of a closest-hit BLAS walk runs in pk_blas_closest_asm (hand-scheduled), 0: compiled walk (A/B)
    int32_t       pk_defer_t0_primary;   // the closest-hit threshold of level 0: the 8x8-pixel packets of primary rays share the top of their walk (pk_blas_closest_asm)
    int32_t       pk_defer_t0_closest;   // the same threshold for closest-hit packets (64 = every lane walks its ray privately from the BLAS root)
    int32_t       pk_defer_t0;    // hybrid any-hit walk: nodes wanted by at most this many lanes (+ walk length / 8) go to the lanes' private stacks
    int32_t       pk_defer_leaf;  // the same for leaves (a leaf costs the packet one triangle test per triangle)
    int32_t       pk_item_cap;    // split walk: items per chunk (one chunk per wave of the shadow-ray packet launch); pk_items == null: off
    uint4 *       pk_items;       // (ray slot, leaf count << 24 | first, instance, -)
    uint32_t *    pk_item_count;  // items in each chunk
    DevCounters * counters;
    uint32_t *    stats_partial;  // [level][stats_stride][4]: partial ray tallies of the level's shading pass, summed by k_resolve(level 0) (no atomics at the end of the pass)
    int32_t       stats_stride;   // entries reserved per level
    int32_t       stats_n[RTX_MAX_LEVELS + 1];     // entries the level's pass writes: k_shade one per workgroup, the fused packet kernel one per wave
    uint32_t *       pk_cost;     // level-0 closest-hit packets: what each packet cost this frame (wave clock ticks >> 6), or null
    const uint32_t * pk_lpt_order;    // ... and the order to take them in: last frame's packets by descending cost (k_packet_order), or null = queue order
    uint32_t *    pk_heads;       // packet-queue heads of the packet kernels: [closest | any][level][8 heads], 128 bytes apart (rtx_packet.h)
    uint32_t *    tm_packed;      // group rendering: level-0 pixels go, packed, to tm_packed[tm_base + primary slot] (tile-major) instead of the raster framebuffer
    int32_t       tm_base;
    float *       fb_rgb;         // W*H*3
    uint32_t *    fb_packed;      // W*H
};
