// rtx_libm.h — transcendental functions of the render path, for the gfx950 kernels.
//
// Where the reference calls libm at lane 1 (SIMD_floats.h:50-59): acosf (Sky.cpp:34, Sphere.cpp:61),
// atan2f (Sphere.cpp:60), expf (Raytracer.cpp:370-372), log2f (Texture.cpp:195,217,272).
//
// acosf / atan2f feed DISCRETE decisions (sky texel index, via truncation) so a 1-ulp difference
// from the host's libm would flip texels.  They are therefore implemented here as the classic
// fdlibm single-precision algorithms (the algorithm glibc 2.35's libm uses for these two
// functions: pure fp32 +,-,*,/ and sqrt, all correctly rounded on gfx950), and are checked
// bit-for-bit against the host libm in tests/test_libm.py (exhaustively for acosf and atanf).
// The file also compiles for the host (plain C++) for exactly that test.
//
// expf / log2f feed continuous quantities (Beer attenuation, mip LOD): they are evaluated in fp64
// and rounded once, i.e. correctly rounded fp32 except in ~1e-9 of cases; glibc's own expf/log2f
// are within 0.502 ulp of that, so the two agree to <= 1 ulp (documented in DESIGN.md).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  #define RTX_LIBM_FN __device__ __forceinline__
  #define RTX_F2I(f) __float_as_int(f)
  #define RTX_I2F(i) __int_as_float(i)
  #define RTX_SQRTF(x) sqrtf(x)
#elif defined(__HIPCC__)
  #define RTX_LIBM_FN __host__ __device__ inline
  #include <math.h>
  #include <string.h>
  static inline int32_t rtx_f2i_host(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
  static inline float rtx_i2f_host(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
  #define RTX_F2I(f) rtx_f2i_host(f)
  #define RTX_I2F(i) rtx_i2f_host(i)
  #define RTX_SQRTF(x) sqrtf(x)
#else
  #define RTX_LIBM_FN static inline
  #include <math.h>
  #include <string.h>
  static inline int32_t rtx_f2i_host(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
  static inline float rtx_i2f_host(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
  #define RTX_F2I(f) rtx_f2i_host(f)
  #define RTX_I2F(i) rtx_i2f_host(i)
  #define RTX_SQRTF(x) sqrtf(x)
#endif

// ---- acosf: fdlibm e_acosf.c ---------------------------------------------------------------------
RTX_LIBM_FN float rtx_acosf(float x) {
    const float one = 1.0f;
    const float pi      = RTX_I2F(0x40490fda);
    const float pio2_hi = RTX_I2F(0x3fc90fda);
    const float pio2_lo = RTX_I2F(0x33a22168);
    const float pS0 = RTX_I2F(0x3e2aaaab), pS1 = RTX_I2F((int32_t)0xbea6b090), pS2 = RTX_I2F(0x3e4e0aa8),
                pS3 = RTX_I2F((int32_t)0xbd241146), pS4 = RTX_I2F(0x3a4f7f04), pS5 = RTX_I2F(0x3811ef08);
    const float qS1 = RTX_I2F((int32_t)0xc019d139), qS2 = RTX_I2F(0x4001572d), qS3 = RTX_I2F((int32_t)0xbf303361), qS4 = RTX_I2F(0x3d9dc62e);
    int32_t hx = RTX_F2I(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) {                       /* |x| == 1 */
        if (hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    } else if (ix > 0x3f800000) {
        return (x - x) / (x - x);                 /* NaN */
    }
    if (ix < 0x3f000000) {                        /* |x| < 0.5 */
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        float z = x * x;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        return pio2_hi - (x - (pio2_lo - r * x));
    } else if (hx < 0) {                          /* x < -0.5 */
        float z = (one + x) * 0.5f;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float s = RTX_SQRTF(z);
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {                                      /* x > 0.5 */
        float z = (one - x) * 0.5f;
        float s = RTX_SQRTF(z);
        float df = RTX_I2F(RTX_F2I(s) & (int32_t)0xfffff000);
        float c = (z - df * df) / (s + df);
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        float w = r * s + c;
        return 2.0f * (df + w);
    }
}

// ---- atanf: fdlibm s_atanf.c ----------------------------------------------------------------------
RTX_LIBM_FN float rtx_atanf(float x) {
    const float one = 1.0f;
    int32_t hx = RTX_F2I(x);
    int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                       /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;        /* NaN */
        const float hi3 = RTX_I2F(0x3fc90fda), lo3 = RTX_I2F(0x33a22168);
        if (hx > 0) return hi3 + lo3;
        return -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {                        /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;            /* |x| < 2^-29 */
        id = -1;
    } else {
        x = RTX_I2F(ix);                          /* fabsf */
        if (ix < 0x3f980000) {                    /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else                 { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    const float aT0 = RTX_I2F(0x3eaaaaab), aT1 = RTX_I2F((int32_t)0xbe4ccccd), aT2 = RTX_I2F(0x3e124925), aT3 = RTX_I2F((int32_t)0xbde38e38),
                aT4 = RTX_I2F(0x3dba2e6e), aT5 = RTX_I2F((int32_t)0xbd9d8795), aT6 = RTX_I2F(0x3d886b35), aT7 = RTX_I2F((int32_t)0xbd6ef16b),
                aT8 = RTX_I2F(0x3d4bda59), aT9 = RTX_I2F((int32_t)0xbd15a221), aT10 = RTX_I2F(0x3c8569d7);
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    float hi, lo;
    if      (id == 0) { hi = RTX_I2F(0x3eed6338); lo = RTX_I2F(0x31ac3769); }
    else if (id == 1) { hi = RTX_I2F(0x3f490fda); lo = RTX_I2F(0x33222168); }
    else if (id == 2) { hi = RTX_I2F(0x3f7b985e); lo = RTX_I2F(0x33140fb4); }
    else              { hi = RTX_I2F(0x3fc90fda); lo = RTX_I2F(0x33a22168); }
    z = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -z : z;
}

// ---- atan2f: fdlibm e_atan2f.c --------------------------------------------------------------------
RTX_LIBM_FN float rtx_atan2f(float y, float x) {
    const float tiny = 1.0e-30f;
    const float pi_o_4 = RTX_I2F(0x3f490fdb), pi_o_2 = RTX_I2F(0x3fc90fdb), pi = RTX_I2F(0x40490fdb), pi_lo = RTX_I2F((int32_t)0xb3bbbd2e);
    int32_t hx = RTX_F2I(x), hy = RTX_F2I(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;       /* NaN */
    if (hx == 0x3f800000) return rtx_atanf(y);                  /* x == 1 */
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else {
        float q = y / x;
        z = rtx_atanf(RTX_I2F(RTX_F2I(q) & 0x7fffffff));
    }
    switch (m) {
        case 0: return z;
        case 1: return RTX_I2F(RTX_F2I(z) ^ (int32_t)0x80000000);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

#if defined(__HIPCC__)
RTX_LIBM_FN float rtx_expf(float x)  { return (float)exp((double)x); }
RTX_LIBM_FN float rtx_log2f(float x) { return (float)log2((double)x); }
#endif
