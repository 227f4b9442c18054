// facade_demo — drives the render path through the reference-shaped C++ facade (rtx_facade.hpp):
//   Scene scene; scene.update(dt); raytracer.render_frame(window, stats);
// usage: facade_demo <scene.rtxs> <width> <height> <bounces> <mip_filter> <texture_mode> [out.ppm]
//        facade_demo --script <scene.txt> <data_dir> <mip_filter> <texture_mode> [out.ppm]     (assets: OBJ/MTL/PNG/TGA -> SBVH -> frame)
// Prints the PerformanceStats and an FNV-1a hash of the packed framebuffer (compared with the Python path in tests).
#include <cstdio>
#include <cstdlib>
#include <string>
#include "rtx_facade.hpp"

// out path ending in .png: the DISPLAYED image (Window::draw_quad: gamma + FXAA) as PNG; anything else: the raw frame as PPM
static bool wants_png(const char * p) { std::string s = p ? p : ""; return s.size() > 4 && s.substr(s.size() - 4) == ".png"; }

static int report(const rtx::Window & window, const rtx::PerformanceStats & stats, const char * ppm) {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t p : window.frame_buffer) for (int k = 0; k < 4; k++) { h ^= (p >> (8 * k)) & 0xff; h *= 1099511628211ull; }
    std::printf("{\"primary\": %lld, \"shadow\": %lld, \"reflection\": %lld, \"refraction\": %lld, \"fnv1a\": \"%016llx\"}\n",
                stats.num_primary_rays, stats.num_shadow_rays, stats.num_reflection_rays, stats.num_refraction_rays, (unsigned long long)h);
    if (ppm && wants_png(ppm)) return rtxh_image_save_png(ppm, window.display.data(), window.width, window.height);
    if (ppm) {
        FILE * f = std::fopen(ppm, "wb");
        if (!f) return 1;
        std::fprintf(f, "P6\n%d %d\n255\n", window.width, window.height);
        for (uint32_t p : window.frame_buffer) { unsigned char c[3] = { (unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p }; std::fwrite(c, 1, 3, f); }
        std::fclose(f);
    }
    return 0;
}

int main(int argc, char ** argv) {
    const bool script = argc >= 6 && std::string(argv[1]) == "--script";
    if (script) {
        try {
            rtx::ScriptHeader h = rtx::peek_script(argv[2]);
            rtx::Config cfg;
            cfg.SCREEN_WIDTH = h.width; cfg.SCREEN_HEIGHT = h.height; cfg.NUMBER_OF_BOUNCES = h.bounces;
            cfg.MIPMAP_FILTER = std::atoi(argv[4]); cfg.TEXTURE_SAMPLE_MODE = std::atoi(argv[5]);
            rtx::Scene scene(cfg, 110.0f * 3.14159265359f * 0.00555555555f /* DEG_TO_RAD(110), Scene.cpp:75, Util.h:14 */);
            scene.USE_BVH_CACHE_FILES = std::getenv("RTX_FACADE_BVH_CACHE") != nullptr;      // like the reference: `<obj>.bvh` next to the meshes
            rtx::load_script(argv[2], argv[3], scene);
            scene.camera.resize(cfg.SCREEN_WIDTH, cfg.SCREEN_HEIGHT);
            scene.upload();
            scene.update(0.0f);
            rtx::Window window(cfg.SCREEN_WIDTH, cfg.SCREEN_HEIGHT);
            rtx::Raytracer raytracer; raytracer.scene = &scene;
            rtx::PerformanceStats stats;
            raytracer.render_frame(window, stats);
            if (argc > 6 && wants_png(argv[6])) raytracer.draw_quad(window);
            return report(window, stats, argc > 6 ? argv[6] : nullptr);
        } catch (const std::exception & e) { std::fprintf(stderr, "facade_demo: %s\n", e.what()); return 1; }
    }
    if (argc < 7) { std::fprintf(stderr, "usage: facade_demo scene.rtxs W H bounces mip_filter texture_mode [out.ppm]\n"); return 2; }
    try {
        rtx::Config cfg;
        cfg.SCREEN_WIDTH = std::atoi(argv[2]); cfg.SCREEN_HEIGHT = std::atoi(argv[3]); cfg.NUMBER_OF_BOUNCES = std::atoi(argv[4]);
        cfg.MIPMAP_FILTER = std::atoi(argv[5]); cfg.TEXTURE_SAMPLE_MODE = std::atoi(argv[6]);
        rtx::Scene scene(cfg, 1.9198622f /* 110 degrees, Scene.cpp:7 */);
        rtx::load_rtxs(argv[1], scene);
        scene.camera.resize(cfg.SCREEN_WIDTH, cfg.SCREEN_HEIGHT);
        scene.upload();
        scene.update(0.0f);                       // instance matrices + TLAS are REBUILT here from position/rotation

        rtx::Window window(cfg.SCREEN_WIDTH, cfg.SCREEN_HEIGHT);
        rtx::Raytracer raytracer; raytracer.scene = &scene;
        rtx::PerformanceStats stats;
        if (std::getenv("RTX_FACADE_GROUP")) {    // the multi-GPU route (rtx_group_*_local) with this one GPU as the whole group
            rtx::RaytracerGroup group; group.init({ &scene });
            group.render_frame(window, stats);
        } else raytracer.render_frame(window, stats);
        if (argc > 7 && wants_png(argv[7])) raytracer.draw_quad(window);
        return report(window, stats, argc > 7 ? argv[7] : nullptr);
    } catch (const std::exception & e) { std::fprintf(stderr, "facade_demo: %s\n", e.what()); return 1; }
    return 0;
}
