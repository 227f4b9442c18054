// rtx_facade.hpp — C++ host facade with the reference's API surface for the render path.
//
// A user of clayne/CPU-Raytracer drives the path through three objects (Main.cpp:39-57):
//     Scene scene;  scene.camera.resize(W, H);  Raytracer raytracer{&scene};
//     loop { scene.update(dt);  for every tile: raytracer.render_tile(window, x, y, w, h, stats);  window.draw_quad(); }
// The classes below keep those names, argument meanings and the Config.h knob names, and forward to the C ABI
// (include/rtx.h on the GPU, include/rtx_host.h for the Scene::update arithmetic).  Scene *content* is data: instead of
// the reference's hard-coded constructors (Scene.cpp:6-131) a Scene is filled from arrays, an RTXS scene file, or asset
// files: Scene::load_mesh / load_texture / add_mesh / add_sphere / add_plane do what BottomLevelBVH::load, Texture::load,
// Mesh::init and the Primitive constructors do in the reference, and load_script reads the scene scripts of
// oracle/ref_harness (the same text drives the real reference in the parity tests).
//
// Header-only; link with librtx_hip.so and librtx_host.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
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
    bool  BVH_VISUALIZE_HEATMAP = false;
    int   device = 0;
    rtx_config c_struct() const {
        rtx_config c; std::memset(&c, 0, sizeof(c));
        c.width = SCREEN_WIDTH; c.height = SCREEN_HEIGHT; c.bounces = NUMBER_OF_BOUNCES; c.stack_size = BVH_TRAVERSAL_STACK_SIZE;
        c.traversal_strategy = BVH_TRAVERSAL_STRATEGY; c.texture_mode = TEXTURE_SAMPLE_MODE; c.mip_filter = MIPMAP_FILTER;
        c.max_anisotropy = MAX_ANISOTROPY; c.device = device; c.heatmap = BVH_VISUALIZE_HEATMAP ? 1 : 0;
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
    // Camera::update(delta, keys), Camera.cpp:18-48: keys = OR of RTXH_KEY_* (what SDL_GetKeyboardState reports in the reference)
    void update(float delta, uint32_t keys) { if (rtxh_camera_update(delta, keys, position, rotation)) throw std::runtime_error("camera"); update(); }
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
    std::vector<uint32_t> display;            // what draw_quad puts on screen (gamma + FXAA), filled by Raytracer::draw_quad
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

    // ---- assets: the reference's global caches (bvh_cache BottomLevelBVH.cpp:16, texture_cache Texture.cpp:11, MaterialBuffer) ----
    int MESH_ACCELERATOR = 1;                                   // Config.h:32-35: 0 = BVH, 1 = SBVH
    int reserve_material() {                                    // MaterialBuffer::reserve with Material's defaults (Material.h:7-14,32-40)
        ensure_default_material();
        rtx_material m; std::memset(&m, 0, sizeof(m));
        m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 1.0f; m.texture_id = -1; m.index_of_refraction = 1.0f;
        materials.push_back(m);
        return (int)materials.size() - 1;
    }
    void load_sky(const std::string & path) {                   // Sky::Sky, Sky.cpp:8-26
        float * tx = nullptr; int32_t n = 0;
        check(rtxh_sky_load(path.c_str(), &tx, &n), ("Sky::Sky " + path).c_str());
        sky.assign(tx, tx + (size_t)n * n * 3); sky_size = n;
        rtxh_texture_free(tx);
    }
    int load_texture(const std::string & path) {                // Texture::load, Texture.cpp:30-129
        auto it = texture_by_file_.find(path);
        if (it != texture_by_file_.end()) return it->second;
        float * tx = nullptr; int64_t count = 0; Texture t;
        check(rtxh_texture_load(path.c_str(), config.TEXTURE_SAMPLE_MODE == RTX_TEXTURE_MIPMAP, &tx, &count, &t.desc), ("Texture::load " + path).c_str());
        t.texels.assign(tx, tx + 3 * count);
        rtxh_texture_free(tx);
        textures.push_back(std::move(t));
        return texture_by_file_[path] = (int)textures.size() - 1;
    }
    bool USE_BVH_CACHE_FILES = false;                           // BottomLevelBVH::load keeps `<obj>.bvh` next to the mesh (BottomLevelBVH.cpp:26-33,53); off by default here
    int load_mesh(const std::string & obj_path) {               // BottomLevelBVH::load, BottomLevelBVH.cpp:18-58
        auto it = blas_by_file_.find(obj_path);
        if (it != blas_by_file_.end()) return it->second;
        ensure_default_material();
        Blas b; b.material_offset = (int)materials.size();      // OBJLoader.cpp:9
        auto take_materials = [&](rtxh_obj & o) {
            for (int i = 0; i < o.mesh.material_count; i++) {
                rtx_material m = o.materials[i];
                m.texture_id = m.texture_id >= 0 ? load_texture(o.texture_names + (size_t)i * RTXH_TEXNAME_MAX) : -1;
                materials.push_back(m);
            }
        };
        const std::string cache_path = obj_path + ".bvh";
        rtxh_bvh_cache cache; std::memset(&cache, 0, sizeof(cache));
        if (USE_BVH_CACHE_FILES && rtxh_bvh_cache_load(cache_path.c_str(), &cache) == RTX_OK) {          // :28-33: cached tree + load_mtl
            rtxh_obj o; std::memset(&o, 0, sizeof(o));
            check(rtxh_mtl_load(obj_path.c_str(), &o), ("OBJLoader::load_mtl " + obj_path).c_str());
            take_materials(o);
            rtxh_obj_free(&o);
            b.nodes.assign(cache.nodes, cache.nodes + cache.node_count);
            b.hot.resize(cache.index_count); b.cold.resize(cache.index_count);                             // flatten, :196-212
            for (int k = 0; k < cache.index_count; k++) { b.hot[k] = cache.hot[cache.indices[k]]; b.cold[k] = cache.cold[cache.indices[k]]; }
            rtxh_bvh_cache_free(&cache);
            blas.push_back(std::move(b));
            return blas_by_file_[obj_path] = (int)blas.size() - 1;
        }
        rtxh_obj o; std::memset(&o, 0, sizeof(o));
        check(rtxh_obj_load(obj_path.c_str(), &o), ("OBJLoader::load_obj " + obj_path).c_str());
        take_materials(o);
        const int n = o.mesh.triangle_count;
        std::vector<int32_t> order; int node_count = 0, refs = n;
        for (int cap = 2 * n;; cap *= 2) {
            b.nodes.assign(2 * (size_t)cap, rtx_bvh_node()); order.assign(cap, 0);
            int rc = MESH_ACCELERATOR == 1
                ? rtxh_blas_build_reference_sbvh(o.mesh.positions, n, b.nodes.data(), (int)b.nodes.size(), &node_count, order.data(), cap, &refs)
                : rtxh_blas_build_reference_bvh(o.mesh.positions, n, b.nodes.data(), &node_count, order.data());
            if (rc == RTX_ERR_LIMIT && MESH_ACCELERATOR == 1 && cap < 32 * n) continue;
            if (rc) { rtxh_obj_free(&o); check(rc, "BottomLevelBVH::build"); }
            break;
        }
        b.nodes.resize(node_count);
        // unflattened arrays first (what the reference writes to the cache, OBJLoader.cpp:156-175), then BottomLevelBVH::flatten :196-212
        std::vector<rtx_triangle_hot> hot(n); std::vector<rtx_triangle_cold> cold(n);
        for (int t = 0; t < n; t++) {
            const float * p = o.mesh.positions + 9 * (size_t)t, * nr = o.mesh.normals + 9 * (size_t)t, * uv = o.mesh.texcoords + 6 * (size_t)t;
            rtx_triangle_hot & h = hot[t]; rtx_triangle_cold & c = cold[t];
            std::memset(&c, 0, sizeof(c));
            for (int d = 0; d < 3; d++) {
                h.position_0[d] = p[d]; h.position_edge_1[d] = p[3 + d] - p[d]; h.position_edge_2[d] = p[6 + d] - p[d];
                c.normal_0[d] = nr[d];  c.normal_edge_1[d] = nr[3 + d] - nr[d]; c.normal_edge_2[d] = nr[6 + d] - nr[d];
            }
            for (int d = 0; d < 2; d++) { c.tex_coord_0[d] = uv[d]; c.tex_coord_edge_1[d] = uv[2 + d] - uv[d]; c.tex_coord_edge_2[d] = uv[4 + d] - uv[d]; }
            c.material_id = o.mesh.material_ids[t];
        }
        rtxh_obj_free(&o);
        if (USE_BVH_CACHE_FILES) {                                                                        // save_to_disk, :53
            rtxh_bvh_cache w; std::memset(&w, 0, sizeof(w));
            w.triangle_count = n; w.node_count = node_count; w.index_count = refs;
            w.hot = hot.data(); w.cold = cold.data(); w.nodes = b.nodes.data(); w.indices = order.data();
            check(rtxh_bvh_cache_save(cache_path.c_str(), &w), ("BottomLevelBVH::save_to_disk " + cache_path).c_str());
        }
        b.hot.resize(refs); b.cold.resize(refs);
        for (int k = 0; k < refs; k++) { b.hot[k] = hot[order[k]]; b.cold[k] = cold[order[k]]; }
        blas.push_back(std::move(b));
        return blas_by_file_[obj_path] = (int)blas.size() - 1;
    }
    Mesh & add_mesh(const std::string & obj_path, const float position[3], const float rotation[4]) {   // Mesh::init, Mesh.cpp:5-7
        Mesh m; m.blas_id = load_mesh(obj_path);
        std::memcpy(m.position, position, 12); std::memcpy(m.rotation, rotation, 16);
        meshes.push_back(m);
        return meshes.back();
    }
    rtx_sphere & add_sphere(const float center[3], float radius) {                                      // Sphere::init, Sphere.h:13-16
        rtx_sphere s; std::memcpy(s.center, center, 12); s.radius_inv = 1.0f / radius; s.radius_squared = radius * radius; s.material_id = reserve_material();
        spheres.push_back(s);
        return spheres.back();
    }
    PlaneDesc & add_plane(const float position[3], const float rotation[4]) {
        PlaneDesc p; std::memcpy(p.position, position, 12); std::memcpy(p.rotation, rotation, 16); p.material_id = reserve_material();
        planes.push_back(p);
        return planes.back();
    }

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
    void ensure_default_material() {                            // MaterialBuffer::init, Material.h:52-60
        if (!materials.empty()) return;
        rtx_material m; std::memset(&m, 0, sizeof(m)); m.texture_id = -1; m.index_of_refraction = 1.0f;
        materials.push_back(m);
    }
    std::map<std::string, int> blas_by_file_, texture_by_file_;
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

    // Window::draw_quad (Window.cpp:87-95) without a GL context: the frame through the reference's gamma / FXAA shaders (rtx_present)
    void draw_quad(Window & window, bool enable_fxaa = true /* ENABLE_FXAA, Config.h:20 */) const {
        window.display.resize((size_t)window.width * window.height);
        scene->check(rtx_present(scene->context(), enable_fxaa ? 1 : 0, window.display.data(), nullptr), "rtx_present");
    }

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

// Several GPUs of one node driven by ONE process — the shape of the reference's own main loop with its worker pool replaced by GPUs
// (WorkerThread.cpp:53-65: tiles handed out round robin; include/rtx.h rtx_group_*).  scenes[i] is the same scene built with
// Config::device = i (every GPU holds the read-only scene); after each scenes[i]->update(delta):
//     group.render_frame(window, stats);        // every GPU renders its tiles, one RCCL gather, rank 0's frame lands in `window`
struct RaytracerGroup {
    std::vector<const Scene *> scenes;
    std::vector<rtx_group *> groups;

    void init(const std::vector<const Scene *> & per_device_scenes) {
        scenes = per_device_scenes;
        std::vector<rtx_ctx *> ctx;
        for (const Scene * s : scenes) ctx.push_back(s->context());
        groups.assign(scenes.size(), nullptr);
        scenes[0]->check(rtx_group_create_local(ctx.data(), (int32_t)ctx.size(), groups.data()), "rtx_group_create_local");
    }
    void render_frame(Window & window, PerformanceStats & stats) const {
        scenes[0]->check(rtx_group_render_local(const_cast<rtx_group **>(groups.data()), (int32_t)groups.size(), 0), "rtx_group_render_local");
        for (const Scene * s : scenes) {                                  // PerformanceStats: summed over the GPUs like WorkerThread.cpp:131-148 over threads
            rtx_stats st;
            s->check(rtx_get_stats(s->context(), &st, nullptr), "rtx_get_stats");
            stats.num_primary_rays += (long long)st.num_primary_rays; stats.num_shadow_rays += (long long)st.num_shadow_rays;
            stats.num_reflection_rays += (long long)st.num_reflection_rays; stats.num_refraction_rays += (long long)st.num_refraction_rays;
        }
        scenes[0]->check(rtx_read_framebuffer(scenes[0]->context(), nullptr, window.frame_buffer.data()), "rtx_read_framebuffer");    // rank 0 holds the assembled packed frame
    }
    ~RaytracerGroup() { for (rtx_group * g : groups) if (g) rtx_group_destroy(g); }
};

// The numbers of the reference's HUD (Main.cpp:59-95): frame time averaged over the last 100 frames, whole-second fps, MRays/s
// = per-frame ray counts x fps x 1e-6.  Call frame(delta_seconds, stats) once per frame, like the body of the game loop.
struct FrameCounters {
    static const int TOTAL_TIMING_COUNT = 100;                 // Main.cpp:15
    float timings[TOTAL_TIMING_COUNT] = { 0 };
    int   current_frame = 0, frames = 0, fps = 0;
    float second = 0.0f, delta_time = 0.0f, avg = 0.0f;
    float mrays_primary = 0, mrays_shadow = 0, mrays_reflection = 0, mrays_refraction = 0, mrays_total = 0;
    void frame(float delta, const PerformanceStats & s) {
        delta_time = delta;
        timings[current_frame++ % TOTAL_TIMING_COUNT] = delta_time;
        avg = 0.0f;
        const int count = current_frame < TOTAL_TIMING_COUNT ? current_frame : TOTAL_TIMING_COUNT;
        for (int i = 0; i < count; i++) avg += timings[i];
        avg /= count;
        frames++;
        second += delta_time;
        while (second >= 1.0f) { second -= 1.0f; fps = frames; frames = 0; }
        mrays_primary = float(s.num_primary_rays * fps) * 1e-6f;       mrays_shadow = float(s.num_shadow_rays * fps) * 1e-6f;
        mrays_reflection = float(s.num_reflection_rays * fps) * 1e-6f; mrays_refraction = float(s.num_refraction_rays * fps) * 1e-6f;
        mrays_total = mrays_primary + mrays_shadow + mrays_reflection + mrays_refraction;
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

// ---- scene scripts (the description language of oracle/ref_harness/refdump.cpp; the same text drives the real reference) --------
struct ScriptHeader { int width = 256, height = 256, bounces = 3; };
inline ScriptHeader peek_script(const std::string & path) {      // size / bounces are needed before the Scene (and its GPU context) exists
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    ScriptHeader h; std::string line;
    while (std::getline(in, line)) {
        std::istringstream is(line.substr(0, line.find('#'))); std::string w; is >> w;
        if (w == "size") is >> h.width >> h.height; else if (w == "bounces") is >> h.bounces;
    }
    return h;
}

// Fills `sc` from a script; `./Data/` prefixes resolve against data_dir (the reference's DATA_PATH, Util.h:6).  Synthetic 64x64 sky as
// in the goldens (the reference's probe files are absent from its repository).
inline void load_script(const std::string & path, const std::string & data_dir, Scene & sc) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    auto data = [&](const std::string & p) { return p.rfind("./Data/", 0) == 0 ? data_dir + "/" + p.substr(7) : p; };
    auto normalize = [](float * v) { float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= inv; v[1] *= inv; v[2] *= inv; };   // Vector3.h:24-31
    const float PI = 3.14159265359f;                              // Util.h:8
    struct Edit { std::string target, field; std::vector<std::string> args; };
    std::vector<Edit> edits;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream is(line.substr(0, line.find('#')));
        std::vector<std::string> w; for (std::string t; is >> t;) w.push_back(t);
        if (w.empty()) continue;
        auto F = [&](size_t i) { return (float)atof(w.at(i).c_str()); };
        auto quat = [&](size_t i, bool axis_angle, bool present, float * q) {
            q[0] = q[1] = q[2] = 0; q[3] = 1;
            if (!present) return;
            if (axis_angle) { float a[3] = { F(i), F(i + 1), F(i + 2) }; rtxh_quaternion_axis_angle(a, F(i + 3), q); }
            else { q[0] = F(i); q[1] = F(i + 1); q[2] = F(i + 2); q[3] = F(i + 3); }
        };
        const std::string & c = w[0];
        if (c == "size" || c == "bounces" || c == "nolights" || c == "texprobe") continue;
        else if (c == "camera" || c == "camera_axis_angle") { sc.camera.position[0] = F(1); sc.camera.position[1] = F(2); sc.camera.position[2] = F(3); quat(4, c == "camera_axis_angle", true, sc.camera.rotation); }
        else if (c == "ambient") { sc.ambient_lighting[0] = F(1); sc.ambient_lighting[1] = F(2); sc.ambient_lighting[2] = F(3); }
        else if (c == "mesh" || c == "mesh_axis_angle") { float p[3] = { F(2), F(3), F(4) }, q[4]; quat(5, c == "mesh_axis_angle", w.size() >= 9, q); sc.add_mesh(data(w[1]), p, q); }
        else if (c == "sphere") { float p[3] = { F(1), F(2), F(3) }; sc.add_sphere(p, F(4)); }
        else if (c == "plane" || c == "plane_axis_angle") { float p[3] = { F(1), F(2), F(3) }, q[4]; quat(4, c == "plane_axis_angle", w.size() >= 8, q); sc.add_plane(p, q); }
        else if (c == "point") { rtx_point_light l = { { F(1), F(2), F(3) }, { F(4), F(5), F(6) } }; sc.point_lights.push_back(l); }
        else if (c == "spot") {                                   // SpotLight.h:12-15
            rtx_spot_light l; float d[3] = { F(7), F(8), F(9) }; normalize(d);
            for (int k = 0; k < 3; k++) { l.colour[k] = F(1 + k); l.position[k] = F(4 + k); l.negative_direction[k] = -d[k]; }
            l.inner_cutoff = cosf((0.5f * F(10)) * PI * 0.00555555555f); l.outer_cutoff = cosf((0.5f * F(11)) * PI * 0.00555555555f);
            sc.spot_lights.push_back(l);
        }
        else if (c == "dir") { rtx_directional_light l; float d[3] = { F(4), F(5), F(6) }; normalize(d); for (int k = 0; k < 3; k++) { l.colour[k] = F(1 + k); l.negative_direction[k] = -d[k]; } sc.directional_lights.push_back(l); }
        else if (c == "matset") { Edit e; e.target = w.at(1); e.field = w.at(2); e.args.assign(w.begin() + 3, w.end()); edits.push_back(e); }
        else throw std::runtime_error("scene script command not supported: " + c);
    }
    for (const Edit & e : edits) {                                // applied after construction, as the harness does
        std::vector<std::string> t; { std::istringstream ts(e.target); for (std::string part; std::getline(ts, part, ':');) t.push_back(part); }
        int id;
        if (t.at(0) == "mesh") id = sc.blas[sc.meshes.at(atoi(t.at(1).c_str())).blas_id].material_offset + atoi(t.at(2).c_str());
        else if (t[0] == "sphere") id = sc.spheres.at(atoi(t.at(1).c_str())).material_id;
        else if (t[0] == "plane") id = sc.planes.at(atoi(t.at(1).c_str())).material_id;
        else if (t[0] == "global") id = atoi(t.at(1).c_str());
        else throw std::runtime_error("bad material target " + e.target);
        rtx_material & m = sc.materials.at(id);
        auto A = [&](size_t i) { return (float)atof(e.args.at(i).c_str()); };
        if (e.field == "diffuse") { m.diffuse[0] = A(0); m.diffuse[1] = A(1); m.diffuse[2] = A(2); }
        else if (e.field == "reflection") { m.reflection[0] = A(0); m.reflection[1] = A(1); m.reflection[2] = A(2); }
        else if (e.field == "transmittance") { m.transmittance[0] = A(0); m.transmittance[1] = A(1); m.transmittance[2] = A(2); }
        else if (e.field == "ior") m.index_of_refraction = A(0);
        else if (e.field == "notexture") m.texture_id = -1;
        else if (e.field == "texture") m.texture_id = sc.load_texture(data(e.args.at(0)));
        else throw std::runtime_error("bad matset field " + e.field);
    }
    const int n = 64;                                             // the goldens' synthetic sky probe (oracle/ref_harness/make_goldens.py)
    sc.sky_size = n; sc.sky.resize((size_t)n * n * 3);
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        float * o = &sc.sky[((size_t)y * n + x) * 3];
        o[0] = (float)(0.5 + 0.5 * x / n); o[1] = (float)(0.6 + 0.2 * (((x / 8) + (y / 8)) & 1)); o[2] = (float)(0.5 + 0.5 * y / n);
    }
}

}  // namespace rtx
