// rtx_texture.h — device-side Texture::sample* and Sky::sample.
// Follows Texture.h:33-49, Texture.cpp:131-337 and Sky.cpp:28-68 operation by operation.
#pragma once
#include "rtx_math.h"
#include "rtx_device.h"
#include "rtx_libm.h"

struct TexCtx {
    const DevTexture * t;
    uint32_t           fetches;    // texel_fetches work counter
};

RTX_D v3 fetch_texel(TexCtx & tc, int x, int y, int level) {        // Texture.cpp:131-147
    const rtx_texture_desc & d = tc.t->desc;
    int offset = d.mip_offsets[level];
    int lw = d.width >> level, lh = d.height >> level;
    x = mod_pos(x, lw);
    y = mod_pos(y, lh);
    tc.fetches++;
    const __attribute__((address_space(1))) float * p = (const __attribute__((address_space(1))) float *)(const void *)tc.t->texels + 3 * (size_t)(offset + x + y * lw);   // texel arrays live in HBM: global_load
    return V3(p[0], p[1], p[2]);
}

RTX_D v3 sample_nearest(TexCtx & tc, float s, float t) {            // Texture.cpp:149-154
    int x = f2i_rn_x86(s * (float)tc.t->desc.width);
    int y = f2i_rn_x86(t * (float)tc.t->desc.height);
    return fetch_texel(tc, x, y, 0);
}

RTX_D v3 sample_bilinear(TexCtx & tc, float s, float t, int level) {  // Texture.cpp:156-186
    int lw = tc.t->desc.width >> level, lh = tc.t->desc.height >> level;
    s = s * (float)lw - 0.5f;
    t = t * (float)lh - 0.5f;
    float fs = s - floorf(s);
    float ft = t - floorf(t);
    float oms = 1.0f - fs, omt = 1.0f - ft;
    float w0 = oms * omt;
    float w1 = fs  * omt;
    float w2 = oms * ft;
    float w3 = 1.0f - w0 - w1 - w2;
    int is = f2i_rn_x86(s - 0.5f);
    int it = f2i_rn_x86(t - 0.5f);
    v3 r =      vmuls(fetch_texel(tc, is,     it,     level), w0);
    r = vadd(r, vmuls(fetch_texel(tc, is + 1, it,     level), w1));
    r = vadd(r, vmuls(fetch_texel(tc, is,     it + 1, level), w2));
    r = vadd(r, vmuls(fetch_texel(tc, is + 1, it + 1, level), w3));
    return r;
}

RTX_D v3 sample_trilinear(TexCtx & tc, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  // Texture.cpp:189-204
    const rtx_texture_desc & d = tc.t->desc;
    float width = 2.0f * std_max(std_max(fabsf(ds_dx), fabsf(ds_dy)), std_max(fabsf(dt_dx), fabsf(dt_dy)));
    float lambda = (float)d.mip_levels - 1.0f + rtx_log2f(std_max(width, 1e-8f));
    int level = f2i_rn_x86(lambda - 0.5f);
    if (level < 0) return sample_bilinear(tc, s, t, 0);
    if (level >= d.mip_levels - 1) return fetch_texel(tc, 0, 0, d.mip_levels - 1);
    float f = lambda - floorf(lambda);
    return vadd(vmuls(sample_bilinear(tc, s, t, level), 1.0f - f), vmuls(sample_bilinear(tc, s, t, level + 1), f));
}

RTX_D v3 sample_anisotropic(TexCtx & tc, float max_aniso, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  // Texture.cpp:207-239
    const rtx_texture_desc & d = tc.t->desc;
    float p_x = std_max(fabsf(ds_dx), fabsf(dt_dx));
    float p_y = std_max(fabsf(ds_dy), fabsf(dt_dy));
    float p_min = std_min(p_x, p_y);
    float p_max = std_max(p_x, p_y);
    float N = std_min(ceilf(p_max / p_min), max_aniso);
    float one_over_N = 1.0f / N;
    float lambda = (float)d.mip_levels - 1.0f + rtx_log2f(p_max * one_over_N);
    int level = f2i_rn_x86(lambda);
    if (level < 0) return sample_bilinear(tc, s, t, 0);
    if (level >= d.mip_levels - 1) return fetch_texel(tc, 0, 0, d.mip_levels - 1);
    bool x_major = p_x > p_y;
    float step_s = x_major ? ds_dx : ds_dy;
    float step_t = x_major ? dt_dx : dt_dy;
    float one_over_N_plus_1 = 1.0f / (N + 1.0f);
    v3 sum = V3(0.0f, 0.0f, 0.0f);
    for (float i = 1.0f; i <= N + 0.001f; i += 1.0f) {
        float x = s + step_s * (i * one_over_N_plus_1 - 0.5f);
        float y = t + step_t * (i * one_over_N_plus_1 - 0.5f);
        sum = vadd(sum, sample_bilinear(tc, x, y, level));
    }
    return vmuls(sum, one_over_N);
}

RTX_D v3 sample_ewa(TexCtx & tc, const float * ewa_table, float max_aniso, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  // Texture.cpp:242-337
    const rtx_texture_desc & d = tc.t->desc;
    float major_x = ds_dx, major_y = dt_dx;
    float minor_x = ds_dy, minor_y = dt_dy;
    float major_length = sqrtf(major_x * major_x + major_y * major_y);
    float minor_length = sqrtf(minor_x * minor_x + minor_y * minor_y);
    if (minor_length > major_length) {
        float tx = minor_x, ty = minor_y, tl = minor_length;
        minor_x = major_x; minor_y = major_y; minor_length = major_length;
        major_x = tx; major_y = ty; major_length = tl;
    }
    if (minor_length < 0.00001f) return sample_bilinear(tc, s, t, 0);
    if (major_length > (float)d.width) return fetch_texel(tc, 0, 0, d.mip_levels - 1);
    if (minor_length * max_aniso < major_length) {
        float scale = major_length / (minor_length * max_aniso);
        minor_x *= scale; minor_y *= scale;
        minor_length *= scale;
    }
    float lambda = std_max(0.0f, (float)d.mip_levels - 1.0f + rtx_log2f(minor_length));
    int level = f2i_rn_x86(lambda);
    if (level >= d.mip_levels - 1) return fetch_texel(tc, 0, 0, d.mip_levels - 1);
    float level_width  = (float)(d.width  >> level);
    float level_height = (float)(d.height >> level);
    s = s * level_width  - 0.5f;
    t = t * level_height - 0.5f;
    float majx = major_x * level_width, majy = major_y * level_height;
    float minx = minor_x * level_width, miny = minor_y * level_height;
    float a =  1.0f + (majy * majy + miny * miny);
    float b = -2.0f * (majx * majy + minx * miny);
    float c =  1.0f + (majx * majx + minx * minx);
    float one_over_f = 1.0f / (a * c - b * b * 0.25f);
    a *= one_over_f; b *= one_over_f; c *= one_over_f;
    float det = -b * b + 4.0f * a * c;
    float sqrt_u = sqrtf(det * c);
    float sqrt_v = sqrtf(det * a);
    float two_inv_det = 2.0f / det;
    float tu = two_inv_det * sqrt_u;
    float tv = two_inv_det * sqrt_v;
    int s0 = f2i_rn_x86(s - tu + 0.5f);
    int s1 = f2i_rn_x86(s + tu - 0.5f);
    int t0 = f2i_rn_x86(t - tv + 0.5f);
    int t1 = f2i_rn_x86(t + tv - 0.5f);
    v3 sum = V3(0.0f, 0.0f, 0.0f);
    float sum_weights = 0.0f;
    float tf = (float)t0;
    const float s0f = (float)s0;
    for (int ti = t0; ti <= t1; ti++, tf += 1.0f) {
        float tt = tf - t;
        float sf = s0f;
        for (int si = s0; si <= s1; si++, sf += 1.0f) {
            float ss = sf - s;
            float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
            if (r2 < 1.0f) {
                int idx = f2i_rn_x86(r2 * (float)RTX_EWA_LUT_SIZE);
                if (RTX_EWA_LUT_SIZE - 1 < idx) idx = RTX_EWA_LUT_SIZE - 1;
                float weight = ewa_table[idx];
                sum = vadd(sum, vmuls(fetch_texel(tc, si, ti, level), weight));
                sum_weights += weight;
            }
        }
    }
    return vdivs(sum, sum_weights);
}

// Texture::sample, Texture.h:33-49 (compile-time switches of Config.h:38-53 become runtime fields)
RTX_D v3 texture_sample(TexCtx & tc, const DevScene & sc, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    if (sc.texture_mode == RTX_TEXTURE_NEAREST)  return sample_nearest(tc, s, t);
    if (sc.texture_mode == RTX_TEXTURE_BILINEAR) return sample_bilinear(tc, s, t, 0);
    if (!tc.t->desc.mipmapped) return sample_bilinear(tc, s, t, 0);
    if (sc.mip_filter == RTX_MIP_TRILINEAR)   return sample_trilinear(tc, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    if (sc.mip_filter == RTX_MIP_ANISOTROPIC) return sample_anisotropic(tc, sc.max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    return sample_ewa(tc, sc.ewa_table, sc.max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
}

// Sky::sample, Sky.cpp:28-68 at lane 1 (truncating float->int, inclusive upper clamp)
RTX_D v3 sky_sample(const float * sky, int size, v3 direction) {
    const float ONE_OVER_PI = 0.31830988618f;
    float r = 0.5f * ONE_OVER_PI * rtx_acosf(direction.z) * (1.0f / sqrtf(direction.x * direction.x + direction.y * direction.y));
    float u = direction.x * r + 0.5f;
    float v = direction.y * r + 0.5f;
    int x = f2i_trunc_x86(u * (float)size);
    int y = f2i_trunc_x86(v * (float)size);
    int index = (int)((uint32_t)y * (uint32_t)size + (uint32_t)x);
    index = index > 0 ? index : 0;
    index = index < size * size ? index : size * size;
    const float * p = sky + 3 * (size_t)index;
    return V3(p[0] * ONE_OVER_PI, p[1] * ONE_OVER_PI, p[2] * ONE_OVER_PI);
}
