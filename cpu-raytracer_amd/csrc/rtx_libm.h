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
// expf / log2f (Beer attenuation, mip LOD — the LOD is floored into a mip level, so it decides discretely too) are glibc 2.35's
// own algorithms: the table-driven double-precision routines of ARM's optimized-routines (exp2f_data N = 32, log2f_data N = 16),
// in the operation order AND fused-multiply-add pattern of the variant glibc selects on every FMA-capable x86-64 CPU
// (`__expf_fma`, `__log2f_fma`, read off the disassembly of libm.so.6; tables and constants extracted from its .rodata).  fp64 +, *
// and fma are correctly rounded on gfx950, so the results are the host's bit for bit (tests/test_libm.py, exhaustive sweep available).
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

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  #define RTX_FMA64(a, b, c) __builtin_fma((a), (b), (c))
  #define RTX_U2D(u) __longlong_as_double((long long)(u))
  #define RTX_D2U(d) ((uint64_t)__double_as_longlong(d))
#else
  static inline double rtx_u2d_host(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
  static inline uint64_t rtx_d2u_host(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
  #define RTX_FMA64(a, b, c) fma((a), (b), (c))
  #define RTX_U2D(u) rtx_u2d_host(u)
  #define RTX_D2U(d) rtx_d2u_host(d)
#endif

// glibc 2.35 __expf_fma (sysdeps/ieee754/flt-32/e_expf.c built with -mfma -mavx2).  T[i] = bits of 2^(i/32) with the exponent field
// pre-adjusted so that adding (ki << 47) yields 2^(k/32) directly.
RTX_LIBM_FN float rtx_expf(float x) {
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
        0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
        0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
    const double SHIFT = 0x1.8p52, InvLn2N = 0x1.71547652b82fep+5, C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const uint32_t ix = (uint32_t)RTX_F2I(x);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop > 0x42a) {                                          // |x| >= 88 or so, inf, NaN
        if (ix == 0xff800000u) return 0.0f;                        // exp(-inf)
        if (abstop > 0x7f7) return x + x;                          // inf, NaN
        if (x > 0x1.62e42ep6f) return RTX_I2F(0x7f800000);         // overflow: __math_oflowf
        if (x < -0x1.9fe368p6f) return 0.0f;                       // underflow: __math_uflowf
        if (x < -0x1.9d1d9ep6f) return RTX_I2F(0x00000001);        // __math_may_uflowf: 0x1.4p-75f squared rounds to the least subnormal
    }
    const double xd = (double)x;
    double kd = RTX_FMA64(InvLn2N, xd, SHIFT);                     // z + SHIFT with z = InvLn2N * xd never rounded on its own
    const uint64_t ki = RTX_D2U(kd);
    kd = kd - SHIFT;
    const double r = RTX_FMA64(InvLn2N, xd, -kd);                  // z - kd
    const double s = RTX_U2D(T[ki & 31] + (ki << 47));
    const double z = RTX_FMA64(r, C0, C1);
    const double r2 = r * r;
    double y = RTX_FMA64(r, C2, 1.0);
    y = RTX_FMA64(z, r2, y);
    y = y * s;
    return (float)y;
}

// glibc 2.35 __log2f_fma (sysdeps/ieee754/flt-32/e_log2f.c built with -mfma -mavx2)
RTX_LIBM_FN float rtx_log2f(float x) {
    const double INVC[16] = { 0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
                              0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                              0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1 };
    const double LOGC[16] = { -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,
                              -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3,
                              0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2 };
    const double A0 = -0x1.712b6f70a7e4dp-2, A1 = 0x1.ecabf496832ep-2, A2 = -0x1.715479ffae3dep-1, A3 = 0x1.715475f35c8b8p+0;
    uint32_t ix = (uint32_t)RTX_F2I(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {           // x < 0x1p-126, inf or NaN
        if (ix * 2u == 0u) return RTX_I2F((int32_t)0xff800000);    // log2(+-0) = -inf (__math_divzerof)
        if (ix == 0x7f800000u) return x;                           // log2(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / (x - x);   // negative or NaN: __math_invalidf
        ix = (uint32_t)RTX_F2I(x * 0x1p23f);                       // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)tmp >> 23;                              // arithmetic shift
    const double z = (double)RTX_I2F((int32_t)iz);
    const double r = RTX_FMA64(z, INVC[i], -1.0);
    const double y0 = LOGC[i] + (double)k;
    const double r2 = r * r;
    double y = RTX_FMA64(A1, r, A2);
    const double p = RTX_FMA64(r, A3, y0);
    y = RTX_FMA64(A0, r2, y);
    y = RTX_FMA64(y, r2, p);
    return (float)y;
}
