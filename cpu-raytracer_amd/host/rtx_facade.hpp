// rtx_facade.hpp — C++ host facade with the reference's API surface for the render path.
//
// A user of clayne/CPU-Raytracer drives the path through three objects (Main.cpp:39-57):
//     Scene scene;  scene.camera.resize(W, H);  Raytracer raytracer{&scene};
//     loop { scene.update(dt);  for every tile: raytracer.render_tile(window, x, y, w, h, stats);  window.draw_quad(); }
// The classes below keep those names, argument meanings and the Config.h knob names, and forward to the C ABI
// (include/rtx.h on the GPU, include/rtx_host.h for the Scene::update arithmetic).  Scene *content* is data: instead of
// the reference's hard-coded constructors (Scene.cpp:6-131) a Scene is filled from arrays or an RTXS scene file.
//
// Header-only; link with librtx_hip.so and librtx_host.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "../../include/rtx_host.h"

namespace rtx {

// Config.h:1-55 — same names, now defaults of a runtime struct
struct Config {
    int   SCREEN_WIDTH = 900, SCREEN_HEIGHT = 600;
    int   NUMBER_OF_BOUNCES = 3;
    int   BVH_TRAVERSAL_STACK_SIZE = 64;
    int   BVH_TRAVERSAL_STRATEGY = RTX_TRAVERSE_ORDERED;
    int   TEXTURE_SAMPLE_MODE = RTX_TEXTURE_MIPMAP;
    int   MIPMAP_FILTER = RTX_MIP_ANISOTROPIC;
    float MAX_ANISOTROPY = 8.0f;
    int   device = 0;
    rtx_config c_struct() const {
        rtx_config c; std::memset(&c, 0, sizeof(c));
        c.width = SCREEN_WIDTH; c.height = SCREEN_HEIGHT; c.bounces = NUMBER_OF_BOUNCES; c.stack_size = BVH_TRAVERSAL_STACK_SIZE;
        c.traversal_strategy = BVH_TRAVERSAL_STRATEGY; c.texture_mode = TEXTURE_SAMPLE_MODE; c.mip_filter = MIPMAP_FILTER;
        c.max_anisotropy = MAX_ANISOTROPY; c.device = device;
        return c;
    }
};

struct PerformanceStats {           // Raytracer.h:4-9
    long long num_primary_rays = 0, num_shadow_rays = 0, num_reflection_rays = 0, num_refraction_rays = 0;
};

struct Camera {                     // Camera.h:10-30
    float position[3] = { 0, 0, 0 };
    float rotation[4] = { 0, 0, 0, 1 };
    float fov;
    int   width = 0, height = 0;
    rtx_camera basis;
    explicit Camera(float fov) : fov(fov) { std::memset(&basis, 0, sizeof(basis)); }
    void resize(int w, int h) { width = w; height = h; }                                   // Camera.cpp:5-16
    bool basis_is_baked = false;   // set when the basis came from a scene file instead of position/rotation/fov
    void update() { if (!basis_is_baked && rtxh_camera_basis(width, height, fov, position, rotation, &basis)) throw std::runtime_error("camera"); }   // :44-47
};

struct Mesh {                       // Mesh.h:8-27 (transform + shared BLAS)
    float position[3] = { 0, 0, 0 };
    float rotation[4] = { 0, 0, 0, 1 };
    int   blas_id = 0;
};

struct Blas {                       // a flattened BottomLevelBVH (BottomLevelBVH.h:5-34)
    std::vector<rtx_bvh_node> nodes; std::vector<rtx_triangle_hot> hot; std::vector<rtx_triangle_cold> cold; int material_offset = 0;
};

struct Texture { rtx_texture_desc desc; std::vector<float> texels; };

class Window {                      // the sink: Window.h:20-65 without SDL / GL
public:
    const int width, height;
    const int tile_width = RTX_TILE_SIZE, tile_height = RTX_TILE_SIZE;
    const int tile_count_x, tile_count_y;
    std::vector<uint32_t> frame_buffer;       // 0x00RRGGBB, as Window::plot packs it
    std::vector<float>    frame_rgb;          // linear fp32 before quantisation
    Window(int w, int h) : width(w), height(h), tile_count_x((w + 31) / 32), tile_count_y((h + 31) / 32),
                           frame_buffer((size_t)w * h), frame_rgb((size_t)w * h * 3) {}
};

class Scene {                       // Scene.h:19-48
public:
    Config config;
    Camera camera;
    std::vector<Blas> blas;
    std::vector<Mesh> meshes;                                   // top_level_bvh.primitives
    std::vector<rtx_sphere> spheres;
    struct PlaneDesc { float position[3]; float rotation[4]; int material_id; };
    std::vector<PlaneDesc> planes;
    std::vector<rtx_plane> baked_planes;                        // already-updated planes from a scene file
    std::vector<rtx_point_light> point_lights;
    std::vector<rtx_spot_light> spot_lights;
    std::vector<rtx_directional_light> directional_lights;
    float ambient_lighting[3] = { 0.2f, 0.2f, 0.2f };           // Scene.h:33
    std::vector<rtx_material> materials;                        // MaterialBuffer::materials
    std::vector<Texture> textures;
    std::vector<float> sky; int sky_size = 0;

    explicit Scene(const Config & cfg, float fov) : config(cfg), camera(fov) {
        rtx_config c = config.c_struct();
        int rc = rtx_create(&c, &ctx_);
        if (rc) throw std::runtime_error("rtx_create failed: " + std::to_string(rc));
    }
    ~Scene() { if (tlas_) rtxh_tlas_destroy(tlas_); if (ctx_) rtx_destroy(ctx_); }
    Scene(const Scene &) = delete; Scene & operator=(const Scene &) = delete;

    // one-off uploads (the reference does this inside its constructors through global caches)
    void upload() {
        for (size_t i = 0; i < blas.size(); i++)
            check(rtx_upload_blas(ctx_, (int)i, blas[i].nodes.data(), (int)blas[i].nodes.size(), blas[i].hot.data(), blas[i].cold.data(),
                                  (int)blas[i].hot.size(), blas[i].material_offset), "rtx_upload_blas");
        check(rtx_upload_materials(ctx_, materials.data(), (int)materials.size()), "rtx_upload_materials");
        for (size_t i = 0; i < textures.size(); i++)
            check(rtx_upload_texture(ctx_, (int)i, &textures[i].desc, textures[i].texels.data(), (int64_t)textures[i].texels.size() / 3), "rtx_upload_texture");
        if (sky_size) check(rtx_upload_sky(ctx_, sky.data(), sky_size), "rtx_upload_sky");
    }

    // Scene::update, Scene.cpp:139-171: camera basis, per-primitive transforms, TLAS rebuild -> one rtx_set_frame
    void update(float /*delta*/) {
        camera.update();
        const int n = (int)meshes.size();
        instances_.resize(n); positions_.resize(3 * (size_t)n); aabbs_.resize(6 * (size_t)n);
        for (int i = 0; i < n; i++) {
            const rtx_bvh_node & root = blas[meshes[i].blas_id].nodes[0];
            std::memset(&instances_[i], 0, sizeof(rtx_instance));
            instances_[i].blas_id = meshes[i].blas_id;
            check(rtxh_instance_update(meshes[i].position, meshes[i].rotation, root.aabb_min, root.aabb_max, &instances_[i], &aabbs_[6 * i], &aabbs_[6 * i + 3]), "instance_update");
            std::memcpy(&positions_[3 * i], meshes[i].position, 12);
        }
        if (n > 0) {
            if (!tlas_ || tlas_n_ != n) { if (tlas_) rtxh_tlas_destroy(tlas_); check(rtxh_tlas_create(n, &tlas_), "tlas_create"); tlas_n_ = n; }
            tlas_nodes_.resize(2 * (size_t)n); tlas_indices_.resize(n);
            check(rtxh_tlas_build(tlas_, positions_.data(), aabbs_.data(), tlas_nodes_.data(), tlas_indices_.data(), &tlas_node_count_), "tlas_build");
        } else tlas_node_count_ = 0;
        planes_.resize(planes.size());
        for (size_t i = 0; i < planes.size(); i++) check(rtxh_plane_update(planes[i].position, planes[i].rotation, planes[i].material_id, &planes_[i]), "plane_update");
        planes_.insert(planes_.end(), baked_planes.begin(), baked_planes.end());

        rtx_frame f; std::memset(&f, 0, sizeof(f));
        f.camera = camera.basis; std::memcpy(f.ambient, ambient_lighting, 12);
        f.tlas_nodes = tlas_nodes_.data(); f.tlas_node_count = tlas_node_count_;
        f.tlas_indices = tlas_indices_.data(); f.tlas_index_count = n;
        f.instances = instances_.data(); f.instance_count = n;
        f.spheres = spheres.data(); f.sphere_count = (int)spheres.size();
        f.planes = planes_.data(); f.plane_count = (int)planes_.size();
        f.point_lights = point_lights.data(); f.point_light_count = (int)point_lights.size();
        f.spot_lights = spot_lights.data(); f.spot_light_count = (int)spot_lights.size();
        f.directional_lights = directional_lights.data(); f.directional_light_count = (int)directional_lights.size();
        check(rtx_set_frame(ctx_, &f), "rtx_set_frame");
    }

    rtx_ctx * context() const { return ctx_; }
    void check(int rc, const char * what) const { if (rc) throw std::runtime_error(std::string(what) + ": " + std::to_string(rc) + " " + rtx_last_error(ctx_)); }

private:
    rtx_ctx * ctx_ = nullptr;
    rtxh_tlas * tlas_ = nullptr; int tlas_n_ = 0; int tlas_node_count_ = 0;
    std::vector<rtx_instance> instances_; std::vector<float> positions_, aabbs_;
    std::vector<rtx_bvh_node> tlas_nodes_; std::vector<int32_t> tlas_indices_; std::vector<rtx_plane> planes_;
};

struct Raytracer {                  // Raytracer.h:11-18
    const Scene * scene = nullptr;

    // Same signature as the reference.  One tile per call works but launches a full kernel pipeline per 32x32 tile;
    // render_frame() below is what replaces the WorkerThread tile loop (WorkerThread.cpp:53-65, 116-129).
    void render_tile(Window & window, int tile_x, int tile_y, int tile_width, int tile_height, PerformanceStats & stats) const {
        (void)tile_width; (void)tile_height;                    // clipped sizes are implied by the tile id (WorkerThread.cpp:60-61)
        const int tile = (tile_y / window.tile_height) * window.tile_count_x + tile_x / window.tile_width;
        render_tiles(window, tile, 1, 1, stats);
    }
    void render_frame(Window & window, PerformanceStats & stats) const { render_tiles(window, 0, 1, window.tile_count_x * window.tile_count_y, stats); }

    void render_tiles(Window & window, int first, int stride, int count, PerformanceStats & stats) const {
        rtx_ctx * c = scene->context();
        scene->check(rtx_render_tiles(c, first, stride, count, 0), "rtx_render_tiles");
        rtx_stats s;
        scene->check(rtx_get_stats(c, &s, nullptr), "rtx_get_stats");
        stats.num_primary_rays += (long long)s.num_primary_rays; stats.num_shadow_rays += (long long)s.num_shadow_rays;
        stats.num_reflection_rays += (long long)s.num_reflection_rays; stats.num_refraction_rays += (long long)s.num_refraction_rays;
        scene->check(rtx_read_framebuffer(c, window.frame_rgb.data(), window.frame_buffer.data()), "rtx_read_framebuffer");
    }
};

// ---- RTXS scene files (this repo's chunk format, see pyrtx/scene_io.py) --------------------------------------------
inline void load_rtxs(const std::string & path, Scene & sc, std::vector<rtx_instance> * raw_instances = nullptr) {
    FILE * f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    char magic[8];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "RTXSCN01", 8)) { std::fclose(f); throw std::runtime_error("not an RTXS file"); }
    std::vector<float> ipos;
    for (;;) {
        char tag[5] = { 0 }; uint32_t n; uint64_t bytes;
        if (std::fread(tag, 1, 4, f) != 4) break;
        if (std::fread(&n, 4, 1, f) != 1 || std::fread(&bytes, 8, 1, f) != 1) break;
        std::vector<char> buf(bytes + 8);
        if (bytes && std::fread(buf.data(), 1, bytes, f) != bytes) break;
        std::fseek(f, (long)((8 - (bytes & 7)) & 7), SEEK_CUR);
        const std::string t(tag);
        auto copy = [&](auto & vec) { using T = typename std::remove_reference<decltype(vec)>::type::value_type; vec.resize(bytes / sizeof(T)); std::memcpy(vec.data(), buf.data(), vec.size() * sizeof(T)); };
        if (t == "CAMR") { std::memcpy(&sc.camera.basis, buf.data(), sizeof(rtx_camera)); std::memcpy(sc.camera.position, buf.data(), 12); sc.camera.basis_is_baked = true; }
        else if (t == "AMBT") std::memcpy(sc.ambient_lighting, buf.data(), 12);
        else if (t == "LPNT") copy(sc.point_lights);
        else if (t == "LSPT") copy(sc.spot_lights);
        else if (t == "LDIR") copy(sc.directional_lights);
        else if (t == "MATL") copy(sc.materials);
        else if (t == "TEXH") { if (sc.textures.size() <= n) sc.textures.resize(n + 1); std::memcpy(&sc.textures[n].desc, buf.data(), sizeof(rtx_texture_desc)); }
        else if (t == "TEXL") { if (sc.textures.size() <= n) sc.textures.resize(n + 1); copy(sc.textures[n].texels); }
        else if (t == "SKYD") { sc.sky_size = (int)n; copy(sc.sky); }
        else if (t == "BLSH") { if (sc.blas.size() <= n) sc.blas.resize(n + 1); sc.blas[n].material_offset = ((int32_t *)buf.data())[2]; }
        else if (t == "BLSN") { if (sc.blas.size() <= n) sc.blas.resize(n + 1); copy(sc.blas[n].nodes); }
        else if (t == "BLST") { if (sc.blas.size() <= n) sc.blas.resize(n + 1); copy(sc.blas[n].hot); }
        else if (t == "BLSC") { if (sc.blas.size() <= n) sc.blas.resize(n + 1); copy(sc.blas[n].cold); }
        else if (t == "INST") { std::vector<rtx_instance> inst; copy(inst); sc.meshes.resize(inst.size()); for (size_t i = 0; i < inst.size(); i++) sc.meshes[i].blas_id = inst[i].blas_id; if (raw_instances) *raw_instances = inst; }
        else if (t == "IPOS") copy(ipos);
        else if (t == "SPHR") copy(sc.spheres);
        else if (t == "PLAN") copy(sc.baked_planes);
    }
    std::fclose(f);
    for (size_t i = 0; i < sc.meshes.size() && 7 * i + 6 < ipos.size(); i++) { std::memcpy(sc.meshes[i].position, &ipos[7 * i], 12); std::memcpy(sc.meshes[i].rotation, &ipos[7 * i + 3], 16); }
}

}  // namespace rtx
