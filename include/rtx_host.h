/*
 * rtx_host.h — C ABI of librtx_host.so: the host-side callers of the render path.
 *
 * CPU-only code that prepares what include/rtx.h consumes, i.e. the reference's per-frame
 * Scene::update (Scene.cpp:139-171) and its load-time BottomLevelBVH construction.  Nothing here
 * runs on the GPU and nothing here renders; the render path itself is librtx_hip.so.
 *
 *   rtxh_camera_basis        Camera::resize + Camera::update basis       Camera.cpp:5-16,44-47
 *   rtxh_camera_update       Camera::update keyboard movement             Camera.cpp:18-39
 *   rtxh_instance_update     Mesh::update                                 Mesh.cpp:9-15
 *                            (Transform::calc_world_matrix Transform.h:13-43, AABB::transform
 *                             AABB.cpp:55-73, Matrix4::invert Matrix4.h:88-138)
 *   rtxh_plane_update        Plane::update                                Plane.cpp:3-11
 *   rtxh_quaternion_axis_angle  Quaternion::axis_angle                     Quaternion.h:26-36
 *   rtxh_scene_update        Scene::update's tail: all Mesh::update + TLAS rebuild   Scene.cpp:166-170
 *   rtxh_scene_dynamic_animate  SCENE_DYNAMIC's animation                   Scene.cpp:141-155
 *   rtxh_tlas_*              TopLevelBVH::init / build_bvh                TopLevelBVH.cpp:5-45
 *                            (BVHBuilders::build_bvh<Mesh> BVHBuilders.h:8-46,
 *                             BVHPartitions::{calculate_bounds,partition_sah,split_indices})
 *   rtxh_blas_build          a BottomLevelBVH for a triangle soup: this repo's OWN binned-SAH
 *                            builder (not the reference's SBVH), output in the
 *                            reference's node convention + flattened leaf order
 *                            (BVHNode.h:10-28, BottomLevelBVH.cpp:196-212)
 *   rtxh_blas_build_reference_bvh   the reference's non-spatial BLAS builder, node for node (BVHBuilders.h:8-46)
 *   rtxh_blas_build_reference_sbvh  the reference's DEFAULT BLAS builder (MESH_ACCELERATOR_SBVH, Config.h:35): spatial
 *                            splits + reference unsplitting, node for node     BVHBuilders.h:48-329, BVHPartitions.h:117-377
 *   rtxh_obj_load            OBJLoader::load_obj / load_mtl                 OBJLoader.cpp:8-187
 *   rtxh_mtl_load            OBJLoader::load_mtl alone                       OBJLoader.cpp:43-68
 *   rtxh_bvh_cache_*         BottomLevelBVH::save_to_disk / load_from_disk   BottomLevelBVH.cpp:149-192
 *   rtxh_texture_mips        Texture::load's box-filter mip chain          Texture.cpp:76-117
 *   rtxh_texture_load        Texture::load: PNG / TGA file -> linear float3 texels + mips   Texture.cpp:30-129
 *   rtxh_image_load          the stbi_load(..., STBI_rgb_alpha) call inside it  Texture.cpp:40
 *   rtxh_sky_load            Sky::Sky: raw float3 angular-map probe file         Sky.cpp:8-26
 *   rtxh_atrium_*            seeded procedural stand-in for the absent Sponza mesh
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include <stdint.h>
#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* position[3], rotation quaternion (x,y,z,w), fov in radians -> rtx_camera */
int rtxh_camera_basis(int32_t width, int32_t height, float fov, const float position[3], const float rotation[4], rtx_camera * out);

/* Mesh::update: fills out->world / world_inv (blas_id untouched) and the world-space AABB of the instance */
int rtxh_instance_update(const float position[3], const float rotation[4], const float blas_root_min[3], const float blas_root_max[3],
                         rtx_instance * out, float out_aabb_min[3], float out_aabb_max[3]);

/* Quaternion::axis_angle (Quaternion.h:26-36) with the host libm's sinf / cosf, as the reference evaluates it */
int rtxh_quaternion_axis_angle(const float axis[3], float angle, float out[4]);

int rtxh_plane_update(const float position[3], const float rotation[4], int32_t material_id, rtx_plane * out);

/* Persistent TLAS builder (keeps the three index arrays across frames like TopLevelBVH does). */
typedef struct rtxh_tlas rtxh_tlas;
int rtxh_tlas_create(int32_t instance_count, rtxh_tlas ** out);
int rtxh_tlas_destroy(rtxh_tlas * t);
/* positions: n*3 (Mesh::get_position = transform.position), aabbs: n*6 (min,max).  nodes_out holds
 * 2*n entries, indices_out n entries; *node_count_out receives TopLevelBVH::node_count.          */
int rtxh_tlas_build(rtxh_tlas * t, const float * positions, const float * aabbs,
                    rtx_bvh_node * nodes_out, int32_t * indices_out, int32_t * node_count_out);

/* Triangle soup -> flattened BLAS.  positions: n*9 floats (p0,p1,p2).  nodes_out must hold 2*n
 * entries, order_out n entries: order_out[k] = source triangle stored at flattened slot k.       */
int rtxh_blas_build(const float * positions, int32_t triangle_count, int32_t bins,
                    rtx_bvh_node * nodes_out, int32_t * node_count_out, int32_t * order_out);

/* The reference's NON-spatial builder (MESH_ACCELERATOR_BVH, BottomLevelBVH::build_bvh BottomLevelBVH.cpp:72-106 over
 * BVHBuilders::build_bvh<Triangle> BVHBuilders.h:8-46): three index lists sorted by triangle centroid
 * (Triangle::get_position, Triangle.h:22-24), full-sweep SAH, leaves below 3 triangles.  Same arguments as rtxh_blas_build;
 * produces the reference's node array and leaf order for that build mode (checked against a reference dump).            */
int rtxh_blas_build_reference_bvh(const float * positions, int32_t triangle_count,
                                  rtx_bvh_node * nodes_out, int32_t * node_count_out, int32_t * order_out);

/* The reference's SBVH builder (MESH_ACCELERATOR_SBVH, the shipped default: BottomLevelBVH::build_sbvh BottomLevelBVH.cpp:108-147 over
 * BVHBuilders::build_sbvh BVHBuilders.h:48-329, BVHPartitions::partition_object / partition_spatial BVHPartitions.h:117-377).
 * A triangle may be referenced from several leaves, so the flattened order is longer than the mesh: order_out receives
 * *order_count_out (>= triangle_count) source-triangle ids, nodes_out *node_count_out nodes.  The reference sizes both
 * arrays at 2 * triangle_count and overruns them beyond that; here a mesh that needs more than the capacities given returns
 * RTX_ERR_LIMIT (call again with larger arrays).  Output checked bit for bit against trees built by the reference.            */
int rtxh_blas_build_reference_sbvh(const float * positions, int32_t triangle_count,
                                   rtx_bvh_node * nodes_out, int32_t node_capacity, int32_t * node_count_out,
                                   int32_t * order_out, int32_t order_capacity, int32_t * order_count_out);

/* Camera::update's input handling (Camera.cpp:18-39): keys = OR of RTXH_KEY_* held during this frame; position / rotation are
 * updated in place (follow with rtxh_camera_basis for the view pyramid, Camera.cpp:44-47).                                      */
enum { RTXH_KEY_W = 1, RTXH_KEY_A = 2, RTXH_KEY_S = 4, RTXH_KEY_D = 8, RTXH_KEY_LSHIFT = 16, RTXH_KEY_SPACE = 32,
       RTXH_KEY_UP = 64, RTXH_KEY_DOWN = 128, RTXH_KEY_LEFT = 256, RTXH_KEY_RIGHT = 512 };
int rtxh_camera_update(float delta, uint32_t keys, float position[3], float rotation[4]);

/* Scene::update in two calls (Scene.cpp:139-171).  rtxh_scene_dynamic_animate is the animation SCENE_DYNAMIC hard-codes (:141-155,
 * instances 0..5; *time is the function-static `time`); rtxh_scene_update is the common tail (:166-170): Mesh::update for every
 * instance, then the TLAS rebuild.  positions n*3, rotations n*4 (x,y,z,w), blas_root_aabbs: 6 floats (min, max) per BLAS id.        */
int rtxh_scene_dynamic_animate(float delta, float * time, float * positions, float * rotations, int32_t instance_count);
int rtxh_scene_update(rtxh_tlas * tlas, int32_t instance_count, const float * positions, const float * rotations, const int32_t * blas_ids,
                      const float * blas_root_aabbs, rtx_instance * instances_out, rtx_bvh_node * tlas_nodes_out,
                      int32_t * tlas_indices_out, int32_t * tlas_node_count_out);

/* Appends the box-filter mip chain to level 0 (texels_rgb holds w*h float3 on entry and must have
 * room for w*h + w*h/3 + 1 texels); fills desc like Texture::load.                               */
int rtxh_texture_mips(float * texels_rgb, int32_t width, int32_t height, rtx_texture_desc * desc, int64_t * texel_count_out);

/* Texture::load (Texture.cpp:30-129): decodes a .png or .tga file the way the reference's vendored stb_image v2.19 does with
 * STBI_rgb_alpha (Texture.cpp:40), converts r,g,b bytes to linear light (colour_unpack :13-20, Math::gamma_to_linear Math.h:67-77;
 * alpha is dropped) and, when mipmap_mode != 0 (TEXTURE_SAMPLE_MODE_MIPMAP) and both sides are powers of two, appends the
 * box-filter chain.  *texels_out is malloc'ed (free with rtxh_texture_free) and holds *texel_count_out float3 texels, ready for
 * rtx_upload_texture.  Returns RTX_ERR_STATE if the file cannot be read, RTX_ERR_LIMIT for an image format stb_image knows
 * but this loader does not (JPEG, BMP, GIF, PSD, PIC, PNM, HDR), RTX_ERR_INVALID_ARG for a corrupt file (the reference aborts). */
int rtxh_texture_load(const char * path, int32_t mipmap_mode, float ** texels_out, int64_t * texel_count_out, rtx_texture_desc * desc);
int rtxh_texture_free(float * texels);
/* Sky::Sky (Sky.cpp:8-26): reads a square angular-map probe stored as raw float3 texels (`Data/Sky_Probes/<name>.float`); the result goes to
 * rtx_upload_sky.  *texels_out is malloc'ed (free with rtxh_texture_free).  A file whose texel count is not a square is refused.     */
int rtxh_sky_load(const char * path, float ** texels_out, int32_t * size_out);
/* The decode step alone: width*height RGBA8 pixels, top row first (malloc'ed; free with rtxh_image_free). */
int rtxh_image_load(const char * path, int32_t * width, int32_t * height, uint8_t ** rgba_out);
int rtxh_image_free(uint8_t * rgba);
/* Screenshot: a frame in Window::frame_buffer layout (0x00RRGGBB, e.g. from rtx_read_framebuffer or rtx_present) as an RGB PNG. */
int rtxh_image_save_png(const char * path, const uint32_t * packed, int32_t width, int32_t height);

/* Procedural "atrium": returns the triangle count for a detail level, then fills caller arrays.   */
typedef struct rtxh_mesh {
    float *   positions;     /* n*9  */
    float *   normals;       /* n*9  */
    float *   texcoords;     /* n*6  (already in the reference's convention: v flipped, OBJLoader.cpp:139-141) */
    int32_t * material_ids;  /* n    (mesh-local) */
    int32_t   triangle_count;
    int32_t   material_count;
} rtxh_mesh;
int rtxh_atrium_generate(uint32_t seed, int32_t detail, rtxh_mesh * out);   /* allocates; free with rtxh_mesh_free */
int rtxh_mesh_free(rtxh_mesh * m);

/* OBJ + MTL -> the triangle soup and material list OBJLoader::load_obj builds (OBJLoader.cpp:8-41,70-187).
 * materials[i].texture_id >= 0 means texture_names + i*RTXH_TEXNAME_MAX holds the map_Kd path (directory of the OBJ
 * prepended, as OBJLoader.cpp:21 does); decoding the image is left to the caller (Texture::load, Texture.cpp:30-129).   */
#define RTXH_TEXNAME_MAX 512
typedef struct rtxh_obj {
    rtxh_mesh      mesh;            /* material_ids are local to `materials` */
    rtx_material * materials;       /* mesh.material_count entries */
    char *         texture_names;   /* mesh.material_count * RTXH_TEXNAME_MAX bytes */
} rtxh_obj;
int rtxh_obj_load(const char * path, rtxh_obj * out);                        /* allocates; free with rtxh_obj_free */
int rtxh_obj_free(rtxh_obj * o);
/* OBJLoader::load_mtl (OBJLoader.cpp:43-68): materials only, from the .mtl named like the OBJ — what BottomLevelBVH::load registers when the
 * geometry comes from a `.bvh` cache file (BottomLevelBVH.cpp:28-33).  out->mesh stays empty except material_count.                          */
int rtxh_mtl_load(const char * obj_path, rtxh_obj * out);

/* `<mesh>.obj.bvh` cache files exactly as BottomLevelBVH::save_to_disk / load_from_disk read and write them (BottomLevelBVH.cpp:149-192).
 * The arrays are NOT flattened: triangle i of a leaf is hot[indices[i]] (BottomLevelBVH::flatten, :196-212, is the caller's gather).          */
typedef struct rtxh_bvh_cache {
    int32_t triangle_count, node_count, index_count, pad;
    rtx_triangle_hot *  hot;       /* triangle_count */
    rtx_triangle_cold * cold;      /* triangle_count */
    rtx_bvh_node *      nodes;     /* node_count */
    int32_t *           indices;   /* index_count */
} rtxh_bvh_cache;
int rtxh_bvh_cache_load(const char * path, rtxh_bvh_cache * out);            /* allocates; free with rtxh_bvh_cache_free */
int rtxh_bvh_cache_save(const char * path, const rtxh_bvh_cache * cache);
int rtxh_bvh_cache_free(rtxh_bvh_cache * cache);

#ifdef __cplusplus
}
#endif
#endif
