// rtx_present.h — the step after the path: Window::draw_quad (Window.cpp:87-95) = the packed frame drawn through
// Data/Shaders/vertex.glsl + fragment_fxaa.glsl (ENABLE_FXAA, Config.h:20) or fragment_identity.glsl, as one HIP kernel.
//
// PARITY UNPINNED against the reference: its shaders need an OpenGL driver, which this environment does not have, so there is
// no reference output to compare with.  The kernel is checked bit for bit against oracle/rt_oracle.c `orc_present`, which
// states the conventions assumed where GL leaves room (fragment centres, nearest / repeat addressing, vec2(float, vec2)
// taking the first component, round-to-nearest-even UNORM8 store, gamma through a 256-entry table).
#pragma once
#include "rtx_device.h"
#include "rtx_math.h"

struct PresentArgs {
    const uint32_t * packed;    // Window::frame_buffer, 0x00RRGGBB (Window.h:56-65)
    const float *    gamma_lut; // pow(i / 255, 1 / 2.2), fragment_*.glsl get_screen
    uint32_t *       display;   // what the colour buffer would hold
    int32_t width, height, fxaa;
};

RTX_D v3 present_screen(const PresentArgs & a, float u, float v) {      // texture2D(screen, uv) with GL_NEAREST / GL_REPEAT + gamma
    const float fu = u - floorf(u), fv = v - floorf(v);
    int x = (int)(fu * (float)a.width), y = (int)(fv * (float)a.height);
    x = x > a.width - 1 ? a.width - 1 : x;
    y = y > a.height - 1 ? a.height - 1 : y;
    const uint32_t p = a.packed[(size_t)y * a.width + x];
    return V3(a.gamma_lut[(p >> 16) & 255], a.gamma_lut[(p >> 8) & 255], a.gamma_lut[p & 255]);
}
RTX_D float present_luma(v3 c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }
RTX_D uint32_t present_unorm8(float c) {
    const float x = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (uint32_t)__float2int_rn(x * 255.0f);
}

__global__ __launch_bounds__(256)
void k_present(const PresentArgs a) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);   // 16x16 pixel blocks: the 3x3 / span taps stay in L1
    if (i >= a.width || j >= a.height) return;
    const float inv_x = 1.0f / (float)a.width, inv_y = 1.0f / (float)a.height;
    const float u = ((float)i + 0.5f) * inv_x, v = ((float)j + 0.5f) * inv_y;
    v3 out;
    if (!a.fxaa) out = present_screen(a, u, v);                                               // fragment_identity.glsl:14-16
    else {                                                                                    // fragment_fxaa.glsl:21-70
        const float luma_tl = present_luma(present_screen(a, u + -inv_x, v + -inv_x));
        const float luma_tr = present_luma(present_screen(a, u +  inv_x, v + -inv_x));
        const float luma_bl = present_luma(present_screen(a, u + -inv_x, v +  inv_x));
        const float luma_br = present_luma(present_screen(a, u +  inv_x, v +  inv_x));
        const float luma_m  = present_luma(present_screen(a, u, v));
        const float luma_min = __builtin_fminf(__builtin_fminf(__builtin_fminf(luma_tl, luma_tr), __builtin_fminf(luma_bl, luma_br)), luma_m);
        const float luma_max = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(luma_tl, luma_tr), __builtin_fmaxf(luma_bl, luma_br)), luma_m);
        float dir_x = (luma_bl + luma_br) - (luma_tl + luma_tr);
        float dir_y = (luma_tl + luma_bl) - (luma_tr + luma_br);
        const float reduce = __builtin_fmaxf(1.0f / 128.0f, (luma_tl + luma_tr + luma_bl + luma_br) * 0.25f * (1.0f / 8.0f));
        const float adjust = 1.0f / (__builtin_fminf(__builtin_fabsf(dir_x), __builtin_fabsf(dir_y)) + reduce);
        dir_x = __builtin_fminf(__builtin_fmaxf(dir_x * adjust, -8.0f), 8.0f) * inv_x;
        dir_y = __builtin_fminf(__builtin_fmaxf(dir_y * adjust, -8.0f), 8.0f) * inv_y;
        const float k1 = 1.0f / 3.0f - 0.5f, k2 = 2.0f / 3.0f - 0.5f, k0 = 0.0f / 3.0f - 0.5f, k3 = 3.0f / 3.0f - 0.5f;
        const v3 a1 = present_screen(a, u + dir_x * k1, v + dir_y * k1), a2 = present_screen(a, u + dir_x * k2, v + dir_y * k2);
        const v3 b1 = present_screen(a, u + dir_x * k0, v + dir_y * k0), b2 = present_screen(a, u + dir_x * k3, v + dir_y * k3);
        const v3 result_a = V3(0.5f * (a1.x + a2.x), 0.5f * (a1.y + a2.y), 0.5f * (a1.z + a2.z));
        const v3 result_b = V3(0.5f * (b1.x + b2.x), 0.5f * (b1.y + b2.y), 0.5f * (b1.z + b2.z));
        const v3 result = V3(0.5f * (result_a.x + result_b.x), 0.5f * (result_a.y + result_b.y), 0.5f * (result_a.z + result_b.z));
        const float luma_result = present_luma(result);
        out = (luma_result < luma_min || luma_result > luma_max) ? result_a : result;
    }
    a.display[(size_t)j * a.width + i] = (present_unorm8(out.x) << 16) | (present_unorm8(out.y) << 8) | present_unorm8(out.z);
}
