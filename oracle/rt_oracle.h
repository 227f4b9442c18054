/*
 * rt_oracle.h — CPU oracle for the render path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (librtx_hip.so and the host facade) never links, loads or calls it.
 *
 * It is a plain-C restatement of the reference's SIMD_LANE_SIZE == 1 path; every function in
 * rt_oracle.c cites the reference file:line it follows.  Pinned: bit-for-bit against the real
 * reference (oracle/_ref/refdump_*, built from /root/reference by oracle/ref_harness/Makefile) on the
 * committed golden scenes under tests/golden/ — see tests/test_oracle_golden.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_blas {
    const rtx_bvh_node *      nodes;
    const rtx_triangle_hot *  tri_hot;
    const rtx_triangle_cold * tri_cold;
    int32_t node_count;
    int32_t triangle_count;
    int32_t material_offset;
    int32_t pad;
} orc_blas;

typedef struct orc_texture {
    rtx_texture_desc desc;
    const float *    texels;       /* float3 per texel, mip chain appended */
} orc_texture;

typedef struct orc_scene {
    rtx_config   config;
    rtx_frame    frame;
    const orc_blas *     blas;       int32_t blas_count;
    const rtx_material * materials;  int32_t material_count;
    const orc_texture *  textures;   int32_t texture_count;
    const float *        sky;        int32_t sky_size;   /* sky must hold sky_size*sky_size + 1 texels */
} orc_scene;

/* Renders tiles first_tile + i*tile_stride (WorkerThread.cpp:57-61 numbering) with `threads`
 * worker threads pulling tiles from an atomic counter (WorkerThread.cpp:53-65).
 * rgb: W*H*3, packed: W*H, dist: W*H (primary hit distance) — each may be NULL.              */
int orc_render_tiles(const orc_scene * scene, int32_t first_tile, int32_t tile_stride, int32_t tile_count,
                     float * rgb, uint32_t * packed, float * dist,
                     rtx_stats * stats, rtx_work_counters * work, int32_t threads);

/* Window::draw_quad (Window.cpp:87-95): the packed frame through the full-screen triangle of Data/Shaders/vertex.glsl and
 * fragment_fxaa.glsl (fxaa != 0, Config.h:20) or fragment_identity.glsl, into an 8-bit display image (0x00RRGGBB).
 * PARITY UNPINNED: a GL driver runs the reference's shaders and none exists here; conventions assumed where GL leaves room are
 * listed at the implementation.  gamma_lut: 256 floats, pow(i / 255, 1 / 2.2) (the caller evaluates it once, with its libm). */
int orc_present(const uint32_t * packed, int32_t width, int32_t height, int32_t fxaa, const float * gamma_lut, uint32_t * display);

/* Unit entry points (each mirrors one reference function). */
int      orc_aabb_intersect(const rtx_bvh_node * node, const float origin[3], const float inv_direction[3], float max_distance);
void     orc_texture_sample(const orc_texture * tex, int32_t texture_mode, int32_t mip_filter, float max_anisotropy,
                            float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy, float out_rgb[3]);
void     orc_sky_sample(const float * sky, int32_t size, const float direction[3], float out_rgb[3]);
uint32_t orc_plot(const float rgb[3]);
void     orc_calc_lighting(const rtx_point_light * pl, const rtx_spot_light * sl, const rtx_directional_light * dl,
                           const float normal[3], const float to_light[3], const float to_camera[3], float distance_squared, float out9[9]);
int32_t  orc_float_to_int(float x);
void     orc_ewa_table(float out[RTX_EWA_LUT_SIZE]);

/* Closest-hit / any-hit of a single ray through the whole scene (Scene::trace_primitives /
 * Scene::intersect_primitives).  ray = origin[3] direction[3] dO_dx[3] dO_dy[3] dD_dx[3] dD_dy[3].
 * hit_out (27 floats): hit, distance, point[3], normal[3], material_id, u, v, ds_dx, ds_dy, dt_dx,
 * dt_dy, dO_dx[3], dO_dy[3], dN_dx[3], dN_dy[3].                                                 */
void orc_trace_closest(const orc_scene * scene, const float ray[18], float hit_out[27]);
int  orc_trace_any(const orc_scene * scene, const float origin[3], const float direction[3], float max_distance);

#ifdef __cplusplus
}
#endif
#endif
