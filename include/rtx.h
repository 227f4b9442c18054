/*
 * rtx.h — C ABI of the MI355X (gfx950) render path.
 *
 * This is the drop-in boundary for the one hot path of clayne/CPU-Raytracer:
 *
 *   Raytracer::render_tile -> Raytracer::bounce -> Scene::trace_primitives /
 *   Scene::intersect_primitives -> TopLevelBVH / Mesh / BottomLevelBVH ->
 *   triangle / Sphere / Plane tests -> Light::calc_lighting -> recursive
 *   reflect / refract with ray differentials -> Texture::sample / Sky::sample
 *   -> Window::plot
 *
 * The reference has no FFI of its own; its seam is three C++ entry points
 * (reference file:line given with each function below).  Every function here
 * takes plain pointers and sizes only, returns an int status (RTX_OK == 0)
 * instead of the reference's printf + abort(), copies caller-owned host
 * buffers before returning, and is implemented by librtx_hip.so (hand-written
 * HIP for gfx950).  There is no CPU fallback behind this ABI: without a HIP
 * device rtx_create() fails with RTX_ERR_NO_DEVICE.
 *
 * Record layouts marked "reference layout" are byte-identical to the
 * reference's lane-1 structs so that its flattened BottomLevelBVH arrays and
 * its `.bvh` cache files (BottomLevelBVH.cpp:149-192) can be handed over
 * without conversion.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 1

/* ---- status codes (the reference aborts; we return) -------------------- */
enum {
    RTX_OK               = 0,
    RTX_ERR_INVALID_ARG  = 1,  /* null pointer, negative count, bad id      */
    RTX_ERR_NO_DEVICE    = 2,  /* no HIP device / device ordinal not present */
    RTX_ERR_HIP          = 3,  /* a HIP runtime call failed (see rtx_last_error) */
    RTX_ERR_LIMIT        = 4,  /* MAX_MATERIALS (Config.h:18), stack size, texture count ... */
    RTX_ERR_STATE        = 5,  /* render before set_frame / missing BLAS ...  */
    RTX_ERR_OOM          = 6
};

/* ---- Config.h knobs (reference Config.h:1-55) --------------------------- */
enum { RTX_TRAVERSE_NAIVE = 0, RTX_TRAVERSE_ORDERED = 1 };            /* Config.h:27-30 */
enum { RTX_TEXTURE_NEAREST = 0, RTX_TEXTURE_BILINEAR = 1, RTX_TEXTURE_MIPMAP = 2 }; /* Config.h:38-42 */
enum { RTX_MIP_TRILINEAR = 0, RTX_MIP_ANISOTROPIC = 1, RTX_MIP_EWA = 2 };           /* Config.h:49-53 */

#define RTX_MAX_MATERIALS   256   /* Config.h:18  MAX_MATERIALS            */
#define RTX_MAX_STACK       64    /* Config.h:25  BVH_TRAVERSAL_STACK_SIZE */
#define RTX_MAX_MIP_LEVELS  16
#define RTX_TILE_SIZE       32    /* Window.h:32-33 tile_width/tile_height */
#define RTX_EWA_LUT_SIZE    128   /* Texture.h:53                          */

typedef struct rtx_config {
    int32_t width;               /* SCREEN_WIDTH                                    */
    int32_t height;              /* SCREEN_HEIGHT                                   */
    int32_t bounces;             /* NUMBER_OF_BOUNCES (0 = primary rays only)       */
    int32_t stack_size;          /* BVH_TRAVERSAL_STACK_SIZE, 1..RTX_MAX_STACK: a BVH with an inner node at depth d needs d + 2 entries of the
                                    reference's (unchecked) per-BVH stack; rtx_render_tiles refuses deeper trees with RTX_ERR_LIMIT */
    int32_t traversal_strategy;  /* BVH_TRAVERSAL_STRATEGY                          */
    int32_t texture_mode;        /* TEXTURE_SAMPLE_MODE                             */
    int32_t mip_filter;          /* MIPMAP_FILTER                                   */
    float   max_anisotropy;      /* MAX_ANISOTROPY                                  */
    int32_t device;              /* HIP device ordinal for this context             */
    int32_t heatmap;             /* BVH_VISUALIZE_HEATMAP (Config.h:23): pixels show BLAS traversal steps of the primary ray (Raytracer.cpp:97-102) */
    int32_t reserved[6];
} rtx_config;

/* ---- geometry records (reference layouts) ------------------------------ */
typedef struct rtx_bvh_node {          /* BVHNode.h:10-16, 32 B */
    float   aabb_min[3];
    float   aabb_max[3];
    int32_t left_or_first;             /* inner: index of left child (right = left+1); leaf: first primitive */
    int32_t count;                     /* bits 30-31: split axis (1=x,2=y,3=z); bits 0-29: leaf count (>0 = leaf) */
} rtx_bvh_node;

typedef struct rtx_triangle_hot {      /* BottomLevelBVH.h:6-10, 36 B */
    float position_0[3];
    float position_edge_1[3];
    float position_edge_2[3];
} rtx_triangle_hot;

typedef struct rtx_triangle_cold {     /* BottomLevelBVH.h:12-22, 64 B */
    float   tex_coord_0[2];
    float   tex_coord_edge_1[2];
    float   tex_coord_edge_2[2];
    float   normal_0[3];
    float   normal_edge_1[3];
    float   normal_edge_2[3];
    int32_t material_id;               /* OBJ-local id; global id = material_offset + this (BottomLevelBVH.cpp:265) */
} rtx_triangle_cold;

typedef struct rtx_material {          /* Material.h:7-14 (texture pointer replaced by a texture id) */
    float   diffuse[3];
    int32_t texture_id;                /* -1 = no texture */
    float   reflection[3];
    float   transmittance[3];
    float   index_of_refraction;
    int32_t pad;
} rtx_material;                        /* 48 B */

typedef struct rtx_texture_desc {      /* Texture.h:8-19 */
    int32_t width;
    int32_t height;
    int32_t mipmapped;                 /* Texture.cpp:49-55: both sides powers of two and mipmap mode */
    int32_t mip_levels;                /* 1 when not mipmapped */
    int32_t mip_offsets[RTX_MAX_MIP_LEVELS]; /* in texels, Texture.cpp:76-117 */
} rtx_texture_desc;

typedef struct rtx_instance {          /* Mesh.h:8-14 after Mesh::update (Mesh.cpp:9-15) */
    int32_t blas_id;
    int32_t pad[3];
    float   world[16];                 /* Transform::world_matrix, cells[i + 4*j] (Matrix4.h:19-23) */
    float   world_inv[16];             /* Mesh::transform_inv */
} rtx_instance;

typedef struct rtx_sphere {            /* Sphere.h:7-24 after Sphere::update */
    float   center[3];                 /* transform.position */
    float   radius_inv;
    float   radius_squared;
    int32_t material_id;               /* global material id (Primitive.h:6) */
} rtx_sphere;

typedef struct rtx_plane {             /* Plane.h:7-22 after Plane::update (Plane.cpp:3-11) */
    float   normal[3];                 /* world_normal */
    float   distance;                  /* world_distance */
    float   u_axis[3];
    float   v_axis[3];
    int32_t material_id;
    int32_t pad;
} rtx_plane;

typedef struct rtx_point_light {       /* PointLight.h:4-7 */
    float colour[3];
    float position[3];
} rtx_point_light;

typedef struct rtx_spot_light {        /* SpotLight.h:6-15 */
    float colour[3];
    float position[3];
    float negative_direction[3];
    float inner_cutoff;                /* cos(half inner angle) */
    float outer_cutoff;                /* cos(half outer angle) */
} rtx_spot_light;

typedef struct rtx_directional_light { /* DirectionalLight.h:4-7 */
    float colour[3];
    float negative_direction[3];
} rtx_directional_light;

typedef struct rtx_camera {            /* Camera.h:10-23 after Camera::update (Camera.cpp:44-47) */
    float position[3];
    float rotated_top_left_corner[3];
    float rotated_x_axis[3];
    float rotated_y_axis[3];
} rtx_camera;

/* Everything Scene::update (Scene.cpp:139-171) produces for one frame. */
typedef struct rtx_frame {
    rtx_camera camera;
    float      ambient[3];             /* Scene::ambient_lighting (Scene.h:33) */
    int32_t    pad0;

    const rtx_bvh_node * tlas_nodes;   /* TopLevelBVH::nodes (root = 0, index 1 unused) */
    int32_t              tlas_node_count;
    const int32_t *      tlas_indices; /* TopLevelBVH::indices: leaf slot -> instance */
    int32_t              tlas_index_count;
    const rtx_instance * instances;
    int32_t              instance_count;

    const rtx_sphere * spheres;                     int32_t sphere_count;
    const rtx_plane *  planes;                      int32_t plane_count;
    const rtx_point_light *       point_lights;       int32_t point_light_count;
    const rtx_spot_light *        spot_lights;        int32_t spot_light_count;
    const rtx_directional_light * directional_lights; int32_t directional_light_count;
} rtx_frame;

/* PerformanceStats (Raytracer.h:4-9), widened to 64 bit; counting rule of
 * Raytracer.cpp:61,163,180,192,265,320 at SIMD_LANE_SIZE == 1.            */
typedef struct rtx_stats {
    uint64_t num_primary_rays;
    uint64_t num_shadow_rays;
    uint64_t num_reflection_rays;
    uint64_t num_refraction_rays;
} rtx_stats;

/* Work counters behind the roofline accounting; filled by
 * rtx_render_tiles only when RTX_RENDER_COUNT_WORK is set (slower kernels). */
typedef struct rtx_work_counters {
    uint64_t closest_rays;             /* rays through trace_primitives          */
    uint64_t any_rays;                 /* rays through intersect_primitives      */
    uint64_t tlas_nodes_closest;       /* TLAS nodes popped (slab tests)         */
    uint64_t tlas_nodes_any;
    uint64_t blas_nodes_closest;       /* BLAS nodes popped (slab tests)         */
    uint64_t blas_nodes_any;
    uint64_t instances_closest;        /* Mesh::trace entered                    */
    uint64_t instances_any;            /* Mesh::intersect entered                */
    uint64_t tri_tests_closest;        /* triangle_trace calls                   */
    uint64_t tri_tests_any;            /* triangle_intersect calls               */
    uint64_t triangle_hits;            /* closest hit is a triangle (cold fetch) */
    uint64_t shaded_hits;              /* material fetches                       */
    uint64_t sky_lookups;
    uint64_t texel_fetches;
    uint64_t rays_spawned;             /* secondary rays written to a queue      */
    uint64_t reserved[5];
} rtx_work_counters;

typedef struct rtx_ctx rtx_ctx;

/* ---- lifetime ----------------------------------------------------------- */

/* One context per GPU.  Replaces: Window(SCREEN_WIDTH, SCREEN_HEIGHT) framebuffer
 * allocation (Window.cpp:42,76), Texture::init() EWA table (Texture.h:53-62,
 * Main.cpp:25), MaterialBuffer::init() (Material.h:52-60, Main.cpp:26) and
 * WorkerThreads::init (WorkerThread.cpp:72-114).                              */
int rtx_create(const rtx_config * config, rtx_ctx ** out_ctx);
int rtx_destroy(rtx_ctx * ctx);
const char * rtx_last_error(const rtx_ctx * ctx);
int rtx_abi_version(void);

/* ---- scene data that lives across frames -------------------------------- */

/* A flattened BottomLevelBVH (BottomLevelBVH.h:5-34 after flatten(),
 * BottomLevelBVH.cpp:196-212): nodes, and hot/cold triangles already in leaf
 * order.  Replaces BottomLevelBVH::load (BottomLevelBVH.cpp:18-59).          */
int rtx_upload_blas(rtx_ctx * ctx, int32_t blas_id,
                    const rtx_bvh_node * nodes, int32_t node_count,
                    const rtx_triangle_hot * tri_hot, const rtx_triangle_cold * tri_cold,
                    int32_t triangle_count, int32_t material_offset);

/* The global MaterialBuffer::materials[] (Material.h:28-61); index 0 is the
 * all-zero default material.                                                 */
int rtx_upload_materials(rtx_ctx * ctx, const rtx_material * materials, int32_t count);

/* One Texture (Texture.h:8-19): linear-space float3 texels, mip chain
 * appended as Texture::load builds it (Texture.cpp:58-117).                  */
int rtx_upload_texture(rtx_ctx * ctx, int32_t texture_id, const rtx_texture_desc * desc,
                       const float * texels_rgb, int64_t texel_count);

/* Sky (Sky.cpp:8-26): size*size float3 texels.                               */
int rtx_upload_sky(rtx_ctx * ctx, const float * texels_rgb, int32_t size);

/* ---- per frame ------------------------------------------------------------ */

/* The result of Scene::update(delta) (Scene.cpp:139-171): camera basis,
 * instance matrices, rebuilt TLAS, analytic primitives, lights.  Must be
 * called before rendering, never concurrently with it (Main.cpp:54-57).      */
int rtx_set_frame(rtx_ctx * ctx, const rtx_frame * frame);

enum {
    RTX_RENDER_COUNT_WORK = 1,         /* also fill rtx_work_counters (instrumented kernels) */
    RTX_RENDER_SIMPLE_TRACE = 2,       /* use the plain pop-and-test traversal kernels (A/B reference for the pair-fetch kernels) */
    RTX_RENDER_SERIAL = 8,             /* launch every kernel on ONE stream: closest-hit + shade per level, then a single shadow-ray
                                          launch for all levels, then resolve.  Use when several contexts render consecutive
                                          frames concurrently (one stream each); the default (two streams, shadow rays of level 0
                                          overlapping the deeper levels) gives the lowest latency for one frame at a time          */
    RTX_RENDER_LANE_TRACE = 16,        /* A/B: round 1's per-lane traversal kernels (persistent threads with lane refill, csrc/rtx_trace.h
                                          k_trace_fast) instead of the hybrid wave-packet kernels (csrc/rtx_packet.h).  Frames are bit-identical. */
    RTX_RENDER_PACKET_STATS = 32,      /* instrumented packet kernels: rtx_work_counters::reserved[0..4] = packets walked, child-pair steps,
                                          triangle steps, and the lane occupancy sums of both                                                 */
    RTX_RENDER_PACKET_CLOSEST = 64,    /* A/B: closest-hit packets walk the shared top of the tree together too (same threshold as shadow-ray packets);
                                          by default every lane of a closest-hit packet walks its own ray from the BLAS root (private stacks in
                                          LDS), which is faster until the shared part of that walk is hand-scheduled like the shadow-ray
                                          kernel's.  Frames are bit-identical either way.                                                   */
    RTX_RENDER_CULL_DEAD_SHADOW_RAYS = 4 /* count, but do not traverse, shadow rays whose unshadowed light contribution is exactly 0
                                          (N.L <= 0 or outside a spot cone): the pixel cannot depend on them.  Frames and
                                          rtx_stats are bit-identical with and without this flag.                                   */
};

/* Renders tiles first_tile + i*tile_stride, i in [0, tile_count), numbered as
 * WorkerThread.cpp:57-61 (task -> x = (task % tile_count_x)*32, y = (task /
 * tile_count_x)*32, clipped at the right/bottom edge), each exactly as
 * Raytracer::render_tile (Raytracer.cpp:3-85) would, into the context's
 * device framebuffer.  Replaces wake_up_worker_threads + wait_on_worker_threads
 * (WorkerThread.cpp:116-129).  Returns after the work has been queued on the
 * context's stream; stats/work (may be NULL) are valid after rtx_synchronize
 * or any read-back call.                                                       */
int rtx_render_tiles(rtx_ctx * ctx, int32_t first_tile, int32_t tile_stride, int32_t tile_count,
                     uint32_t flags);
int rtx_synchronize(rtx_ctx * ctx);
int rtx_get_stats(rtx_ctx * ctx, rtx_stats * stats, rtx_work_counters * work);

/* Window::plot sink (Window.h:56-65).  rgb_f32: width*height*3 floats (linear,
 * before quantisation); packed_u32: width*height 0x00RRGGBB words.  Either
 * may be NULL.  Pixels of tiles that were never rendered are zero.            */
int rtx_read_framebuffer(rtx_ctx * ctx, float * rgb_f32, uint32_t * packed_u32);

/* Window::draw_quad (Window.cpp:87-95): what the window would show.  The packed frame is drawn through the reference's
 * full-screen shaders — gamma 1/2.2 and, with enable_fxaa (ENABLE_FXAA, Config.h:20), Data/Shaders/fragment_fxaa.glsl — by one
 * HIP kernel queued behind the frame.  display_u32 (host, width*height 0x00RRGGBB, may be NULL) receives the image after a
 * stream synchronise; *display_dev (may be NULL) its device address.  Parity with the GL path is UNPINNED (no GL driver to run
 * the reference's shaders); the kernel equals oracle `orc_present` bit for bit, which documents the conventions assumed.        */
int rtx_present(rtx_ctx * ctx, int32_t enable_fxaa, uint32_t * display_u32, void ** display_dev);

/* Device addresses of the same two buffers (for RCCL / zero-copy consumers). */
int rtx_framebuffer_device_ptrs(rtx_ctx * ctx, void ** rgb_f32_dev, void ** packed_u32_dev);

/* Render into caller-owned device buffers (e.g. tensors of a framework that also owns the RCCL
 * communicator) instead of the context's own framebuffer; NULL restores the internal one.  The
 * buffers must hold width*height*3 floats and width*height uint32.                                */
int rtx_bind_framebuffer(rtx_ctx * ctx, void * rgb_f32_dev, void * packed_u32_dev);

/* Enqueue all work of this context on a caller-owned HIP stream (hipStream_t passed as void *);
 * NULL restores the context's own stream.  Lets a caller order rendering against its own
 * kernels / collectives without host synchronisation.                                              */
int rtx_set_stream(rtx_ctx * ctx, void * hip_stream);

/* Timing of every kernel launched since rtx_enable_kernel_timing(ctx, 1),
 * measured with HIP events on the stream the kernels are launched on.
 * names/ms hold up to `capacity` entries; *count receives the number of
 * kernel launches recorded.                                                    */
int rtx_last_kernel_times(rtx_ctx * ctx, const char ** names, float * ms, int32_t capacity, int32_t * count);
int rtx_enable_kernel_timing(rtx_ctx * ctx, int32_t enable);

/* ---- several GPUs of one node (BASELINE configs[3]) ------------------------------------------------------
 * The reference hands 32x32 screen tiles to worker threads through an atomic counter (WorkerThread.cpp:53-65,
 * 116-129); tiles are independent given the read-only Scene.  A GPU group does the same across GPUs: tile t
 * belongs to rank t mod world, every rank renders its tiles into a TILE-MAJOR buffer of packed pixels (edge
 * tiles padded to 32x32), ONE RCCL gather per frame moves those buffers to rank 0 over xGMI, and rank 0 writes
 * them into its packed framebuffer (rtx_read_framebuffer / rtx_framebuffer_device_ptrs on rank 0's context).
 * Every rank uploads the same scene and calls rtx_set_frame with the same frame (Scene::update runs on every
 * host, or its result is sent: < 4 KiB).  All calls of a group are collective: every rank makes them in the
 * same order.  RCCL (librccl.so.1) is loaded at run time; without it these functions return RTX_ERR_STATE.   */
typedef struct rtx_group rtx_group;
#define RTX_GROUP_ID_BYTES 128
/* rank 0: a fresh communicator id (ncclGetUniqueId) to hand to the other ranks by any out-of-band channel */
int rtx_group_unique_id(void * id128);
/* one context per process: rank `rank` of `world` joins the group (id128 may be NULL when world == 1) */
int rtx_group_create(rtx_ctx * ctx, int32_t rank, int32_t world, const void * id128, rtx_group ** out_group);
/* this rank's share of one frame + gather + (rank 0) frame assembly; queued on the context's stream, returns at once.
 * flags as rtx_render_tiles (RTX_RENDER_SERIAL is implied).  rtx_get_stats afterwards covers this rank's tiles. */
int rtx_group_render(rtx_group * group, uint32_t flags);
int rtx_group_destroy(rtx_group * group);
/* several frames in flight on one GPU (one context each, as bench.py runs them): the further contexts JOIN the first one's communicator
 * as the same rank instead of opening their own.  The gathers of all members are then issued on one exchange stream per rank in call
 * order (events tie them to the frames' own streams), so every rank issues its collectives in the same order whatever order its frames
 * finish in, and a rank never has two collectives in flight.  Every rank must call rtx_group_render on its members in the same order. */
int rtx_group_attach(rtx_ctx * ctx, rtx_group * base, rtx_group ** out_group);
/* one process driving n GPUs, contexts[i] on its own device = rank i (ncclCommInitAll); render issues all ranks' calls */
int rtx_group_create_local(rtx_ctx ** contexts, int32_t n, rtx_group ** out_groups);
int rtx_group_render_local(rtx_group ** groups, int32_t n, uint32_t flags);
/* The partition and the tile-major slot order as plain host functions (no GPU, no context): the same code the kernels run.
 * rank `rank` of `world` owns tiles rank, rank + world, ... (numbering of WorkerThread.cpp:57-61: the reference hands those task
 * numbers to its threads through an atomic counter, :53-65); every rank sends tiles_per_rank tiles of 1024 packed pixels.
 * rtx_group_slot_pixels: pixel_index[i] = y * width + x of slot i of the rank's send buffer, -1 for padding; rank 0 receives the
 * send buffers concatenated in rank order.  capacity >= tiles_per_rank * 1024. */
int rtx_group_layout(int32_t width, int32_t height, int32_t world, int32_t rank, int32_t * tiles_total, int32_t * tiles_per_rank, int32_t * own_tiles);
int rtx_group_slot_pixels(int32_t width, int32_t height, int32_t world, int32_t rank, int64_t * pixel_index, int64_t capacity);

/* ---- unit-level entry points: one reference function each, evaluated on the device -------------
 * (used by the parity tests; not needed by a renderer)                                          */

/* fn: 0 acosf(a) 1 atan2f(a,b) 2 expf(a) 3 log2f(a) 4 atanf(a) 5 Util::float_to_int(a) 6 1/sqrtf(a) */
int rtx_debug_libm(rtx_ctx * ctx, int32_t fn, const float * a, const float * b, float * out, int32_t n);
/* Texture::sample (Texture.h:33-49) of an uploaded texture at n inputs (s,t,ds_dx,ds_dy,dt_dx,dt_dy) */
int rtx_debug_texture_sample(rtx_ctx * ctx, int32_t texture_id, const float * in6, float * out_rgb, int32_t n);
/* Scene::trace_primitives (Scene.cpp:173-180) of n rays (18 floats each: origin, direction, dO_dx, dO_dy, dD_dx, dD_dy) through the
 * production closest-hit kernel + the accept-branch rebuild of the shade kernel; out: 27 floats per ray = every RayHit field
 * (RayHit.h:5-21): hit, distance, point[3], normal[3], material_id, u, v, ds_dx, ds_dy, dt_dx, dt_dy, dO_dx[3], dO_dy[3], dN_dx[3], dN_dy[3]
 * (zero on a miss except distance).  flags: RTX_RENDER_LANE_TRACE / RTX_RENDER_PACKET_CLOSEST select the kernel.  Needs bounces >= 1. */
int rtx_debug_trace_rays(rtx_ctx * ctx, const float * rays18, int32_t n, float * hits27, uint32_t flags);
/* Scene::intersect_primitives (Scene.cpp:182-190) of n rays (7 floats: origin, direction, max distance) through the production shadow-ray kernel */
int rtx_debug_occluded(rtx_ctx * ctx, const float * origin_direction_maxdist7, int32_t n, uint32_t * occluded, uint32_t flags);
/* Point / Spot / DirectionalLight::calc_lighting (10 floats in: normal, to_light, to_camera, distance^2; 9 out) and Window::plot */
int rtx_debug_light_plot(rtx_ctx * ctx, const rtx_point_light * pl, const rtx_spot_light * sl, const rtx_directional_light * dl,
                         const float * in10, float * out9, int32_t n_light, const float * rgb, uint32_t * packed, int32_t n_plot);
/* the group path of `world` ranks replayed on this one GPU without RCCL (partition, tile-major writes, frame assembly) */
int rtx_debug_group_loopback(rtx_ctx * ctx, int32_t world, uint32_t flags);
/* the shadow-ray packet walk of an uploaded mesh: *stack_need = packet-stack entries its 4-wide records can need, or -1 when the mesh keeps
 * the binary walk (boxes not nested / limits exceeded) */
int rtx_debug_blas_wide(rtx_ctx * ctx, int32_t blas_id, int32_t * stack_need);
/* the per-lane part of the closest-hit packet walk of an uploaded mesh: *stack_need = stack entries a lane's walk of the ordered 4-wide records
 * can need, or -1 when the mesh keeps the binary walk (boxes not nested / a leaf of 16+ triangles / 2^24+ nodes or triangles) */
int rtx_debug_blas_wide_closest(rtx_ctx * ctx, int32_t blas_id, int32_t * stack_need);
/* Sky::sample (Sky.cpp:28-68) of the uploaded sky at n directions */
int rtx_debug_sky_sample(rtx_ctx * ctx, const float * directions_xyz, float * out_rgb, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
