// GpuBackend.cpp — drop-in replacement for WorkerThread.cpp in the clayne/CPU-Raytracer tree: the same four WorkerThreads entry points
// (WorkerThread.h:8-14), implemented on include/rtx.h instead of a Win32 thread pool.  Main.cpp's loop (Main.cpp:51-57) does not change:
//
//     scene.update(delta);                                  // Scene.cpp:139-171, unchanged, on the host
//     WorkerThreads::wake_up_worker_threads(tile_count);    // -> rtx_set_frame + rtx_render_tiles
//     window.clear(); WorkerThreads::wait_on_worker_threads();   // -> rtx_read_framebuffer into Window::frame_buffer
//     window.draw_quad();
//
// Build: compile this file INSTEAD of WorkerThread.cpp, add <repo>/include to the include path, link librtx_hip.so.
// Five reference classes keep the data the GPU needs in private members (Texture texels, Sky texels, Sphere radii, Plane axes,
// Window::frame_buffer); a maintainer grants this file access with one `friend struct GpuBackendAccess;` line in each of Texture.h:9,
// Sky.h:5, Sphere.h:8, Plane.h:8, Window.h:20.  `make -C oracle/ref_harness binding_check` compiles this very file against the
// reference's headers where they lie (read-only there, so the check lifts access control with -Dprivate=public instead).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "WorkerThread.h"
#include "Material.h"
#include "rtx.h"

struct GpuBackendAccess {                                          // the friend named above
    static const Vector3 * texels(const Texture & t) { return t.data; }
    static void describe(const Texture & t, rtx_texture_desc & d, int64_t & texel_count) {
        memset(&d, 0, sizeof(d));
        d.width = t.width; d.height = t.height; d.mipmapped = t.mipmapped ? 1 : 0; d.mip_levels = t.mipmapped ? t.mip_levels : 1;
        texel_count = (int64_t)t.width * t.height;
        if (t.mipmapped) {
            for (int l = 0; l < t.mip_levels && l < RTX_MAX_MIP_LEVELS; l++) d.mip_offsets[l] = t.mip_offsets[l];
            const int lw = t.width >> (t.mip_levels - 1), lh = t.height >> (t.mip_levels - 1);      // extent Texture::load wrote, Texture.cpp:76-117
            texel_count = t.mip_offsets[t.mip_levels - 1] + (int64_t)(lw > 0 ? lw : 1) * (lh > 0 ? lh : 1);
        }
    }
    static int sky_size(const Sky & s) { return s.size; }
    static const Vector3 * sky_texels(const Sky & s) { return s.data; }
    static void sphere(const Sphere & s, rtx_sphere & o) {
        o.center[0] = s.transform.position.x; o.center[1] = s.transform.position.y; o.center[2] = s.transform.position.z;
        o.radius_inv = s.radius_inv; o.radius_squared = s.radius_squared; o.material_id = s.material_id;
    }
    static void plane(const Plane & p, rtx_plane & o) {
        memset(&o, 0, sizeof(o));
        o.normal[0] = p.world_normal.x; o.normal[1] = p.world_normal.y; o.normal[2] = p.world_normal.z; o.distance = p.world_distance;
        o.u_axis[0] = p.u_axis.x; o.u_axis[1] = p.u_axis.y; o.u_axis[2] = p.u_axis.z;
        o.v_axis[0] = p.v_axis.x; o.v_axis[1] = p.v_axis.y; o.v_axis[2] = p.v_axis.z;
        o.material_id = p.material_id;
    }
    static unsigned * frame_buffer(const Window & w) { return w.frame_buffer; }
};

static_assert(sizeof(BVHNode) == sizeof(rtx_bvh_node), "BVHNode.h:10-16 layout is the ABI's");
static_assert(sizeof(BottomLevelBVH::TriangleHot) == sizeof(rtx_triangle_hot) && sizeof(BottomLevelBVH::TriangleCold) == sizeof(rtx_triangle_cold), "BottomLevelBVH.h:6-22");

namespace {
rtx_ctx * ctx = nullptr;
const Raytracer * raytracer = nullptr;
const Window * window = nullptr;
std::vector<const BottomLevelBVH *> blas_table;                    // blas id = index of the BottomLevelBVH in order of first use
std::vector<const Texture *> texture_table;

void check(int rc, const char * what) {                            // the reference's own error style: printf + abort (e.g. Material.h:33-37)
    if (rc != RTX_OK) { printf("%s failed: %d %s\n", what, rc, ctx ? rtx_last_error(ctx) : ""); abort(); }
}
void put(float * d, const Vector3 & v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
void put(float * d, const SIMD_Vector3 & v) { d[0] = v.x[0]; d[1] = v.y[0]; d[2] = v.z[0]; }     // lane 0 of the broadcast value
int blas_id_of(const BottomLevelBVH * b) {
    for (size_t i = 0; i < blas_table.size(); i++) if (blas_table[i] == b) return (int)i;
    blas_table.push_back(b);
    return (int)blas_table.size() - 1;
}
int texture_id_of(const Texture * t) {
    if (!t) return -1;
    for (size_t i = 0; i < texture_table.size(); i++) if (texture_table[i] == t) return (int)i;
    texture_table.push_back(t);
    return (int)texture_table.size() - 1;
}

// MaterialBuffer::materials (Material.h:28-61) -> rtx_material[], every Texture they name -> rtx_upload_texture, Scene::sky -> rtx_upload_sky
void upload_materials_textures_sky(const Scene & scene) {
    std::vector<rtx_material> mats(MaterialBuffer::material_count);
    for (int i = 0; i < MaterialBuffer::material_count; i++) {
        const Material & m = MaterialBuffer::materials[i];
        memset(&mats[i], 0, sizeof(rtx_material));
        put(mats[i].diffuse, m.diffuse); put(mats[i].reflection, m.reflection); put(mats[i].transmittance, m.transmittance);
        mats[i].index_of_refraction = m.index_of_refraction;
        mats[i].texture_id = texture_id_of(m.texture);
    }
    for (size_t t = 0; t < texture_table.size(); t++) {
        rtx_texture_desc d; int64_t n = 0;
        GpuBackendAccess::describe(*texture_table[t], d, n);
        check(rtx_upload_texture(ctx, (int32_t)t, &d, &GpuBackendAccess::texels(*texture_table[t])->x, n), "rtx_upload_texture");
    }
    check(rtx_upload_materials(ctx, mats.data(), (int32_t)mats.size()), "rtx_upload_materials");
    check(rtx_upload_sky(ctx, &GpuBackendAccess::sky_texels(scene.sky)->x, GpuBackendAccess::sky_size(scene.sky)), "rtx_upload_sky");
}

// everything Scene::update (Scene.cpp:139-171) produced for this frame
void set_frame_from_scene(const Scene & s) {
    rtx_frame f; memset(&f, 0, sizeof(f));
    put(f.camera.position, s.camera.position);
    put(f.camera.rotated_top_left_corner, s.camera.rotated_top_left_corner);
    put(f.camera.rotated_x_axis, s.camera.rotated_x_axis);
    put(f.camera.rotated_y_axis, s.camera.rotated_y_axis);
    put(f.ambient, s.ambient_lighting);
    const TopLevelBVH & tl = s.top_level_bvh;
    std::vector<rtx_instance> inst(tl.primitive_count);
    for (int i = 0; i < tl.primitive_count; i++) {
        const Mesh & m = tl.primitives[i];
        memset(&inst[i], 0, sizeof(rtx_instance));
        inst[i].blas_id = blas_id_of(m.bvh);
        memcpy(inst[i].world, &m.transform.world_matrix, 64);      // Matrix4 = 16 floats, cells[i + 4 j] (Matrix4.h:19-23)
        memcpy(inst[i].world_inv, &m.transform_inv, 64);
    }
    std::vector<rtx_sphere> spheres(s.spheres.primitive_count);
    for (int i = 0; i < s.spheres.primitive_count; i++) GpuBackendAccess::sphere(s.spheres.primitives[i], spheres[i]);
    std::vector<rtx_plane> planes(s.planes.primitive_count);
    for (int i = 0; i < s.planes.primitive_count; i++) GpuBackendAccess::plane(s.planes.primitives[i], planes[i]);
    std::vector<rtx_point_light> pl(s.point_light_count);
    for (int i = 0; i < s.point_light_count; i++) { put(pl[i].colour, s.point_lights[i].colour); put(pl[i].position, s.point_lights[i].position); }
    std::vector<rtx_spot_light> sl(s.spot_light_count);
    for (int i = 0; i < s.spot_light_count; i++) {
        const SpotLight & l = s.spot_lights[i];
        put(sl[i].colour, l.colour); put(sl[i].position, l.position); put(sl[i].negative_direction, l.negative_direction);
        sl[i].inner_cutoff = l.inner_cutoff; sl[i].outer_cutoff = l.outer_cutoff;
    }
    std::vector<rtx_directional_light> dl(s.directional_light_count);
    for (int i = 0; i < s.directional_light_count; i++) { put(dl[i].colour, s.directional_lights[i].colour); put(dl[i].negative_direction, s.directional_lights[i].negative_direction); }
    f.tlas_nodes = (const rtx_bvh_node *)tl.nodes; f.tlas_node_count = tl.node_count;
    f.tlas_indices = tl.indices; f.tlas_index_count = tl.primitive_count;
    f.instances = inst.data(); f.instance_count = (int32_t)inst.size();
    f.spheres = spheres.data(); f.sphere_count = (int32_t)spheres.size();
    f.planes = planes.data(); f.plane_count = (int32_t)planes.size();
    f.point_lights = pl.data(); f.point_light_count = (int32_t)pl.size();
    f.spot_lights = sl.data(); f.spot_light_count = (int32_t)sl.size();
    f.directional_lights = dl.data(); f.directional_light_count = (int32_t)dl.size();
    check(rtx_set_frame(ctx, &f), "rtx_set_frame");                // copies everything before it returns
}
}   // namespace

void WorkerThreads::init(const Raytracer & rt, const Window & win) {
    raytracer = &rt; window = &win;
    rtx_config c; memset(&c, 0, sizeof(c));
    c.width = win.width; c.height = win.height; c.bounces = NUMBER_OF_BOUNCES; c.stack_size = BVH_TRAVERSAL_STACK_SIZE;
    c.traversal_strategy = BVH_TRAVERSAL_STRATEGY; c.texture_mode = TEXTURE_SAMPLE_MODE; c.mip_filter = MIPMAP_FILTER;
    c.max_anisotropy = MAX_ANISOTROPY; c.heatmap = BVH_VISUALIZE_HEATMAP ? 1 : 0; c.device = 0;
    check(rtx_create(&c, &ctx), "rtx_create");
    const Scene & s = *rt.scene;
    for (int i = 0; i < s.top_level_bvh.primitive_count; i++) {    // every distinct BottomLevelBVH once (BottomLevelBVH::load caches per file, :16-33)
        const BottomLevelBVH * b = s.top_level_bvh.primitives[i].bvh;
        const int before = (int)blas_table.size(), id = blas_id_of(b);
        if (id == before)
            check(rtx_upload_blas(ctx, id, (const rtx_bvh_node *)b->nodes, b->node_count, (const rtx_triangle_hot *)b->triangles_hot,
                                  (const rtx_triangle_cold *)b->triangles_cold, b->index_count, b->material_offset), "rtx_upload_blas");
    }
    upload_materials_textures_sky(s);
}

void WorkerThreads::wake_up_worker_threads(int job_count) {        // Main.cpp:56, right after scene.update(delta)
    set_frame_from_scene(*raytracer->scene);
    check(rtx_render_tiles(ctx, 0, 1, job_count, 0), "rtx_render_tiles");     // job_count == tile_count_x * tile_count_y, numbered as WorkerThread.cpp:57-61
}

void WorkerThreads::wait_on_worker_threads() {                     // Main.cpp:59-61: the frame must be in Window::frame_buffer for draw_quad
    check(rtx_read_framebuffer(ctx, nullptr, GpuBackendAccess::frame_buffer(*window)), "rtx_read_framebuffer");
}

PerformanceStats WorkerThreads::sum_performance_stats() {          // WorkerThread.cpp:131-148 returns ray slots (packets x SIMD_LANE_SIZE): rays here
    rtx_stats s; check(rtx_get_stats(ctx, &s, nullptr), "rtx_get_stats");
    PerformanceStats p;
    p.num_primary_rays = (int)s.num_primary_rays; p.num_shadow_rays = (int)s.num_shadow_rays;
    p.num_reflection_rays = (int)s.num_reflection_rays; p.num_refraction_rays = (int)s.num_refraction_rays;
    return p;
}
