// rtx_texture.h — texture and sky look-ups of the shade kernel (the reference's Texture::sample, Texture.h:33-49, and Sky::sample,
// Sky.cpp:28-68, at SIMD_LANE_SIZE 1).
//
// Layout and fetch path (MI355X):
//   * a texel is ONE float4 in HBM (rgb + padding, converted at rtx_upload_texture): one global_load_dwordx4 per texel instead of
//     three dword loads, 16-byte aligned so a 2x2 footprint is four independent 16-byte requests;
//   * a mip level is described once per sample by a TexLevel (base pointer, extent, wrap masks).  Mip chains only exist for
//     power-of-two textures (Texture.cpp:49-55), so their wrap-around is an AND with (extent - 1) — identical to Math::mod
//     (Math.h:54-61) for every int, two's complement — instead of two integer divisions per texel; other extents keep the modulo;
//   * a bilinear footprint wraps each of its two columns / rows once, issues its four loads together and only then blends;
//     the anisotropic filter fetches the footprints of two taps at a time (eight loads in flight) before it accumulates them.
//
// What is pinned by parity and therefore kept exactly: every fp32 operation and its order in the level-of-detail, weight and
// coordinate arithmetic and in the accumulation of texels (the probes of tests/test_gpu_units.py come from the real reference and a
// different association flips a mip level or a last bit); float -> int is cvtss2si (Util.h:39-41).
#pragma once
#include "rtx_math.h"
#include "rtx_device.h"
#include "rtx_libm.h"

struct TexLevel {                       // one mip level of one texture, as the samplers see it
    rtx_gptr base;                      // first texel of the level (float4 each)
    int w, h;
    int mask_x, mask_y;                 // extent - 1 when the extent is a power of two, else -1 (use the modulo)
    float wf, hf;
};

struct TexCtx {
    const DevTexture * t;
    uint32_t           fetches;         // texel_fetches work counter (COUNT builds)
};

RTX_D TexLevel tex_level(const TexCtx & tc, int level) {
    const rtx_texture_desc & d = tc.t->desc;
    TexLevel L;
    L.w = d.width >> level; L.h = d.height >> level;
    L.base = RTX_GPTR(tc.t->texels) + d.mip_offsets[level];
    L.mask_x = (L.w & (L.w - 1)) == 0 ? L.w - 1 : -1;
    L.mask_y = (L.h & (L.h - 1)) == 0 ? L.h - 1 : -1;
    L.wf = (float)L.w; L.hf = (float)L.h;
    return L;
}
RTX_D int tex_wrap(int v, int extent, int mask) { return mask >= 0 ? (v & mask) : mod_pos(v, extent); }      // Math::mod, Texture.cpp:139-140
RTX_D v3 tex_load(const TexLevel & L, int x, int y) { const float4 c = gld(L.base, x + y * L.w); return V3(c.x, c.y, c.z); }

// Texture::fetch_texel, Texture.cpp:131-147
RTX_D v3 tex_texel(TexCtx & tc, const TexLevel & L, int x, int y) {
    tc.fetches++;
    return tex_load(L, tex_wrap(x, L.w, L.mask_x), tex_wrap(y, L.h, L.mask_y));
}

// A bilinear footprint (Texture::sample_bilinear, Texture.cpp:156-186): where the four texels are and how they are weighted.
struct TexQuad { int x0, x1, y0, y1; float w00, w10, w01, w11; };
RTX_D TexQuad tex_quad(const TexLevel & L, float s, float t) {
    const float ps = s * L.wf - 0.5f, pt = t * L.hf - 0.5f;
    const float fs = ps - floorf(ps), ft = pt - floorf(pt);
    const float gs = 1.0f - fs, gt = 1.0f - ft;
    TexQuad q;
    q.w00 = gs * gt; q.w10 = fs * gt; q.w01 = gs * ft;
    q.w11 = 1.0f - q.w00 - q.w10 - q.w01;
    const int ix = f2i_rn_x86(ps - 0.5f), iy = f2i_rn_x86(pt - 0.5f);
    q.x0 = tex_wrap(ix, L.w, L.mask_x); q.x1 = tex_wrap(ix + 1, L.w, L.mask_x);
    q.y0 = tex_wrap(iy, L.h, L.mask_y); q.y1 = tex_wrap(iy + 1, L.h, L.mask_y);
    return q;
}
struct TexQuadData { v3 c00, c10, c01, c11; };
RTX_D TexQuadData tex_quad_fetch(TexCtx & tc, const TexLevel & L, const TexQuad & q) {      // four independent 16-byte loads
    TexQuadData d;
    d.c00 = tex_load(L, q.x0, q.y0); d.c10 = tex_load(L, q.x1, q.y0); d.c01 = tex_load(L, q.x0, q.y1); d.c11 = tex_load(L, q.x1, q.y1);
    tc.fetches += 4;
    return d;
}
RTX_D v3 tex_quad_blend(const TexQuad & q, const TexQuadData & d) {
    v3 r = vmuls(d.c00, q.w00);
    r = vadd(r, vmuls(d.c10, q.w10));
    r = vadd(r, vmuls(d.c01, q.w01));
    r = vadd(r, vmuls(d.c11, q.w11));
    return r;
}
RTX_D v3 tex_bilinear(TexCtx & tc, const TexLevel & L, float s, float t) {
    const TexQuad q = tex_quad(L, s, t);
    return tex_quad_blend(q, tex_quad_fetch(tc, L, q));
}

// Texture::sample_nearest, Texture.cpp:149-154
RTX_D v3 tex_nearest(TexCtx & tc, float s, float t) {
    const TexLevel L = tex_level(tc, 0);
    return tex_texel(tc, L, f2i_rn_x86(s * L.wf), f2i_rn_x86(t * L.hf));
}

// the 1x1 top of the chain, which every mip filter returns beyond its last level
RTX_D v3 tex_coarsest(TexCtx & tc) { const TexLevel L = tex_level(tc, tc.t->desc.mip_levels - 1); return tex_texel(tc, L, 0, 0); }

// Texture::sample_mipmap_trilinear, Texture.cpp:189-204: both levels' footprints are fetched before either is blended
RTX_D v3 tex_trilinear(TexCtx & tc, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    const int levels = tc.t->desc.mip_levels;
    const float footprint = 2.0f * std_max(std_max(fabsf(ds_dx), fabsf(ds_dy)), std_max(fabsf(dt_dx), fabsf(dt_dy)));
    const float lod = (float)levels - 1.0f + rtx_log2f(std_max(footprint, 1e-8f));
    const int lo = f2i_rn_x86(lod - 0.5f);
    if (lo < 0) return tex_bilinear(tc, tex_level(tc, 0), s, t);
    if (lo >= levels - 1) return tex_coarsest(tc);
    const TexLevel La = tex_level(tc, lo), Lb = tex_level(tc, lo + 1);
    const TexQuad qa = tex_quad(La, s, t), qb = tex_quad(Lb, s, t);
    const TexQuadData da = tex_quad_fetch(tc, La, qa), db = tex_quad_fetch(tc, Lb, qb);
    const float frac = lod - floorf(lod);
    return vadd(vmuls(tex_quad_blend(qa, da), 1.0f - frac), vmuls(tex_quad_blend(qb, db), frac));
}

// Texture::sample_mipmap_anisotropic, Texture.cpp:207-239: up to MAX_ANISOTROPY bilinear taps along the longer screen-space axis of
// the footprint, on the level chosen by the shorter one.  Taps are fetched two at a time; they are ADDED in tap order.
RTX_D v3 tex_anisotropic(TexCtx & tc, float max_taps, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    const int levels = tc.t->desc.mip_levels;
    const float ext_x = std_max(fabsf(ds_dx), fabsf(dt_dx)), ext_y = std_max(fabsf(ds_dy), fabsf(dt_dy));
    const float ext_short = std_min(ext_x, ext_y), ext_long = std_max(ext_x, ext_y);
    const float taps = std_min(ceilf(ext_long / ext_short), max_taps);
    const float inv_taps = 1.0f / taps;
    const float lod = (float)levels - 1.0f + rtx_log2f(ext_long * inv_taps);
    const int level = f2i_rn_x86(lod);
    if (level < 0) return tex_bilinear(tc, tex_level(tc, 0), s, t);
    if (level >= levels - 1) return tex_coarsest(tc);
    const TexLevel L = tex_level(tc, level);
    const bool along_x = ext_x > ext_y;
    const float axis_s = along_x ? ds_dx : ds_dy, axis_t = along_x ? dt_dx : dt_dy;
    const float spacing = 1.0f / (taps + 1.0f);
    const float last = taps + 0.001f;
    v3 acc = V3(0.0f, 0.0f, 0.0f);
    for (float k = 1.0f; k <= last; k += 2.0f) {
        const float ka = k, kb = k + 1.0f;                    // the tap counter advances by repeated += 1.0f in the reference: exact for these small integers
        const bool second = kb <= last;
        const float offa = ka * spacing - 0.5f, offb = kb * spacing - 0.5f;
        const TexQuad qa = tex_quad(L, s + axis_s * offa, t + axis_t * offa);
        const TexQuad qb = tex_quad(L, s + axis_s * offb, t + axis_t * offb);
        const TexQuadData da = tex_quad_fetch(tc, L, qa);
        TexQuadData db = da;
        if (second) db = tex_quad_fetch(tc, L, qb);
        acc = vadd(acc, tex_quad_blend(qa, da));
        if (second) acc = vadd(acc, tex_quad_blend(qb, db));
    }
    return vmuls(acc, inv_taps);
}

// Texture::sample_mipmap_ewa, Texture.cpp:242-337: elliptical weighted average over the texels inside the unit ellipse of the pixel's
// footprint, Gaussian weights from the 128-entry table of Texture::init (Texture.h:53-62).  Texels are visited row by row (a row's
// wrap is computed once) and accumulated in that order.
RTX_D v3 tex_ewa(TexCtx & tc, const float * lut, float max_ratio, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    const rtx_texture_desc & d = tc.t->desc;
    float ax = ds_dx, ay = dt_dx, bx = ds_dy, by = dt_dy;                    // the two axes of the footprint; a becomes the major one
    float alen = sqrtf(ax * ax + ay * ay), blen = sqrtf(bx * bx + by * by);
    if (blen > alen) { const float x = bx, y = by, l = blen; bx = ax; by = ay; blen = alen; ax = x; ay = y; alen = l; }
    if (blen < 0.00001f) return tex_bilinear(tc, tex_level(tc, 0), s, t);
    if (alen > (float)d.width) return tex_coarsest(tc);
    if (blen * max_ratio < alen) { const float k = alen / (blen * max_ratio); bx *= k; by *= k; blen *= k; }      // clamp the eccentricity
    const float lod = std_max(0.0f, (float)d.mip_levels - 1.0f + rtx_log2f(blen));
    const int level = f2i_rn_x86(lod);
    if (level >= d.mip_levels - 1) return tex_coarsest(tc);
    const TexLevel L = tex_level(tc, level);
    const float cs = s * L.wf - 0.5f, ct = t * L.hf - 0.5f;                 // ellipse centre in texels
    const float Ax = ax * L.wf, Ay = ay * L.hf, Bx = bx * L.wf, By = by * L.hf;
    float qa =  1.0f + (Ay * Ay + By * By);                                  // implicit ellipse qa*u^2 + qb*u*v + qc*v^2 < 1
    float qb = -2.0f * (Ax * Ay + Bx * By);
    float qc =  1.0f + (Ax * Ax + Bx * Bx);
    const float norm = 1.0f / (qa * qc - qb * qb * 0.25f);
    qa *= norm; qb *= norm; qc *= norm;
    const float det = -qb * qb + 4.0f * qa * qc;
    const float half_w = (2.0f / det) * sqrtf(det * qc), half_h = (2.0f / det) * sqrtf(det * qa);      // bounding box of the ellipse
    const int x_lo = f2i_rn_x86(cs - half_w + 0.5f), x_hi = f2i_rn_x86(cs + half_w - 0.5f);
    const int y_lo = f2i_rn_x86(ct - half_h + 0.5f), y_hi = f2i_rn_x86(ct + half_h - 0.5f);
    v3 acc = V3(0.0f, 0.0f, 0.0f);
    float wsum = 0.0f;
    float yf = (float)y_lo;
    const float xf0 = (float)x_lo;
    for (int y = y_lo; y <= y_hi; y++, yf += 1.0f) {
        const float v = yf - ct;
        const int row = tex_wrap(y, L.h, L.mask_y) * L.w;
        float xf = xf0;
        for (int x = x_lo; x <= x_hi; x++, xf += 1.0f) {
            const float u = xf - cs;
            const float r2 = qa * u * u + qb * u * v + qc * v * v;
            if (r2 < 1.0f) {
                int slot = f2i_rn_x86(r2 * (float)RTX_EWA_LUT_SIZE);
                if (RTX_EWA_LUT_SIZE - 1 < slot) slot = RTX_EWA_LUT_SIZE - 1;
                const float w = lut[slot];
                const float4 c = gld(L.base, tex_wrap(x, L.w, L.mask_x) + row);
                tc.fetches++;
                acc = vadd(acc, vmuls(V3(c.x, c.y, c.z), w));
                wsum += w;
            }
        }
    }
    return vdivs(acc, wsum);
}

// Texture::sample, Texture.h:33-49 (the compile-time switches of Config.h:38-53 are run-time fields of the context)
RTX_D v3 texture_sample(TexCtx & tc, const DevScene & sc, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    if (sc.texture_mode == RTX_TEXTURE_NEAREST)  return tex_nearest(tc, s, t);
    if (sc.texture_mode == RTX_TEXTURE_BILINEAR || !tc.t->desc.mipmapped) return tex_bilinear(tc, tex_level(tc, 0), s, t);
    if (sc.mip_filter == RTX_MIP_TRILINEAR)   return tex_trilinear(tc, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    if (sc.mip_filter == RTX_MIP_ANISOTROPIC) return tex_anisotropic(tc, sc.max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    return tex_ewa(tc, sc.ewa_table, sc.max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
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
