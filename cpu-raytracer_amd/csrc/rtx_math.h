// rtx_math.h — fp32 helpers for the gfx950 kernels.
//
// Every helper reproduces the operation ORDER of the reference's SIMD_LANE_SIZE == 1 path, because
// pixel parity needs bit-identical hit/miss decisions (a 1-ulp change in a slab or triangle test
// flips a silhouette pixel).  The translation unit is compiled with -ffp-contract=off and HIP's
// correctly rounded fp32 divide / sqrt, so `a * b + c` below is an unfused multiply and add exactly
// like the reference's lane-1 madd (SIMD_floats.h:47-48).
//
//   dot(l,r)  = l.x*r.x + (l.y*r.y + l.z*r.z)            SIMD_Vector3.h:53-55
//   cross     = msub pattern                              SIMD_Vector3.h:57-63
//   v / f     = v * (1.0f / f)                            SIMD_Vector3.h:137,153
//   min/max   = a<b?a:b / a>b?a:b (NaN -> b)              SIMD_floats.h:32-33
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTX_D __device__ __forceinline__

struct v3 { float x, y, z; };

RTX_D v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_D v3 v3p(const float * p) { return V3(p[0], p[1], p[2]); }
RTX_D v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_D v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_D v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
RTX_D v3 vmuls(v3 a, float f) { return V3(a.x * f, a.y * f, a.z * f); }
RTX_D v3 vadds(v3 a, float f) { return V3(a.x + f, a.y + f, a.z + f); }
RTX_D v3 vneg(v3 a) { return V3(0.0f - a.x, 0.0f - a.y, 0.0f - a.z); }          // SIMD_Vector3.h:133-136
RTX_D float vdot(v3 l, v3 r) { return l.x * r.x + (l.y * r.y + l.z * r.z); }
RTX_D v3 vcross(v3 l, v3 r) { return V3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x); }
RTX_D v3 vdivs(v3 a, float f) { float inv = 1.0f / f; return vmuls(a, inv); }
RTX_D v3 vnormalize(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vmuls(a, inv); }
RTX_D v3 vmadd_s(v3 a, float b, v3 c) { return V3(a.x * b + c.x, a.y * b + c.y, a.z * b + c.z); }
RTX_D v3 vmadd_v(v3 a, v3 b, v3 c) { return V3(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z); }
RTX_D float fmin_ref(float a, float b) { return a < b ? a : b; }
RTX_D float fmax_ref(float a, float b) { return a > b ? a : b; }
RTX_D float std_max(float a, float b) { return (a < b) ? b : a; }                 // std::max
RTX_D float std_min(float a, float b) { return (b < a) ? b : a; }                 // std::min
RTX_D v3 vrcp(v3 a) { return V3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }            // SIMD_Vector3::rcp

// Util::float_to_int = cvtss2si (Util.h:39-41): round to nearest even, "integer indefinite"
// 0x80000000 for NaN / out of range (v_cvt_i32_f32 would saturate instead).
RTX_D int f2i_rn_x86(float x) { return (fabsf(x) < 2147483648.0f) ? __float2int_rn(x) : (int)0x80000000; }
// SIMD_float_to_int at lane 1 (SIMD.h:11) = C truncation = cvttss2si.
RTX_D int f2i_trunc_x86(float x) { return (fabsf(x) < 2147483648.0f) ? (int)x : (int)0x80000000; }

// Matrix4::transform_position / transform_direction (SIMD versions), Matrix4.h:39-85; cells[i + 4j]
RTX_D v3 xform_pos(const float * c, v3 d) {
    return V3(c[0] * d.x + (c[1] * d.y + (c[2]  * d.z + c[3])),
              c[4] * d.x + (c[5] * d.y + (c[6]  * d.z + c[7])),
              c[8] * d.x + (c[9] * d.y + (c[10] * d.z + c[11])));
}
RTX_D v3 xform_dir(const float * c, v3 d) {
    return V3(c[0] * d.x + (c[1] * d.y + c[2]  * d.z),
              c[4] * d.x + (c[5] * d.y + c[6]  * d.z),
              c[8] * d.x + (c[9] * d.y + c[10] * d.z));
}

RTX_D float pow2_128(float v) {   // Math::pow2<128>, Math.h:92-96: seven squarings
    v = v * v; v = v * v; v = v * v; v = v * v; v = v * v; v = v * v; v = v * v;
    return v;
}

RTX_D int mod_pos(int value, int modulus) {   // Math::mod, Math.h:54-61
    int r = value % modulus;
    if (r < 0) r += modulus;
    return r;
}

RTX_D float as_f(int i) { return __int_as_float(i); }
RTX_D int   as_i(float f) { return __float_as_int(f); }
