// Micro-benchmark behind bench.py's roofline: the chip's ceiling for the ARITHMETIC of the traversal, with every operand already in
// registers (no memory access, no control flow, no stack):
//   * AABB::intersect (AABB.cpp:38-52) as the packet kernels evaluate it: 6 packed sub/mul + 6 min/max + min3/max3 + clamp + compare
//     = 17 VALU instructions per box test and lane;
//   * triangle_intersect (BottomLevelBVH.cpp:311-346) with the correctly rounded division: 62 VALU instructions per test and lane.
// A traversal kernel cannot run faster than (slab tests x cost + triangle tests x cost) / this rate: that is the roofline `peak` the
// benchmark line quotes (unit: G slab-test equivalents per second for the whole chip); tools/roofline_peak.sh runs it and stores the
// output under profiles/.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// boxes live in SGPRs (kernel arguments), rays in VGPRs; 8 independent box tests per iteration so the VALU never waits on a dependency
__global__ __launch_bounds__(256) void slab_loop(const float4 b0, const float4 b1, int iters, float * out) {
    const float l = (float)(threadIdx.x & 63);
    f2v oxy = { l * 0.01f, l * 0.02f }, ozz = { l * 0.03f, l * 0.03f }, ixy = { 1.0f + l * 0.001f, 1.1f }, izz = { 0.9f, 0.9f };
    float tcur = 100.0f;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            const f2v t0xy = (f2v{ b0.x + (float)k, b0.y } - oxy) * ixy;
            const f2v t1xy = (f2v{ b0.z + (float)k, b0.w } - oxy) * ixy;
            const f2v tz = (f2v{ b1.x, b1.y + (float)k } - ozz) * izz;
            const float tn = vmax(0.005f, vmax3(vmin(t0xy.x, t1xy.x), vmin(t0xy.y, t1xy.y), vmin(tz.x, tz.y)));
            const float tf = vmin(tcur, vmin3(vmax(t0xy.x, t1xy.x), vmax(t0xy.y, t1xy.y), vmax(tz.x, tz.y)));
            acc += __ballot(tn < tf);
        }
        oxy.x += 1e-6f; ozz += 1e-6f; ixy.y += 1e-6f; izz -= 1e-6f;      // every term of the test depends on the iteration
    }
    if (acc == 0x1234567ull) out[0] = 1.0f;
}

// one triangle in SGPRs, the ray in VGPRs; 2 independent triangle tests per iteration
__global__ __launch_bounds__(256) void tri_loop(const float4 p0, const float4 e1, const float4 e2, int iters, float * out) {
    const float l = (float)(threadIdx.x & 63);
    float ox = l * 0.01f, oy = l * 0.02f, oz = -3.0f, dx = 0.01f * l, dy = 0.02f, dz = 1.0f, tcur = 100.0f;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int k = 0; k < 2; k++) {
            const float e2x = e2.x + (float)k, e1x = e1.x;
            const float hx = dy * e2.z - dz * e2.y, hy = dz * e2x - dx * e2.z, hz = dx * e2.y - dy * e2x;
            const float a = e1x * hx + (e1.y * hy + e1.z * hz);
            const float f = 1.0f / a;
            const float sx = ox - p0.x, sy = oy - p0.y, sz = oz - p0.z;
            const float u = f * (sx * hx + (sy * hy + sz * hz));
            const float qx = sy * e1.z - sz * e1.y, qy = sz * e1x - sx * e1.z, qz = sx * e1.y - sy * e1x;
            const float v = f * (dx * qx + (dy * qy + dz * qz));
            const float t = f * (e2x * qx + (e2.y * qy + e2.z * qz));
            acc += __ballot((u > 0.0f) & (u < 1.0f) & (v > 0.0f) & ((u + v) < 1.0f) & (t > 0.005f) & (t < tcur));
        }
        ox += 1e-6f; dx += 1e-6f; dy -= 1e-6f; dz += 2e-6f;      // every term of the test depends on the iteration: nothing is loop-invariant
    }
    if (acc == 0x1234567ull) out[0] = 1.0f;
}

int main() {
    float * out; CHECK(hipMalloc(&out, 4));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 20000;
    for (int wps : { 2, 4, 8 }) {
        const int blocks = cus * wps;
        float best_s = 1e30f, best_t = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            float ms;
            CHECK(hipEventRecord(a)); hipLaunchKernelGGL(slab_loop, dim3(blocks), dim3(256), 0, 0, make_float4(-1, -1, 1, 1), make_float4(-1, 1, 0, 0), iters, out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best_s) best_s = ms;
            CHECK(hipEventRecord(a)); hipLaunchKernelGGL(tri_loop, dim3(blocks), dim3(256), 0, 0, make_float4(0, 0, 0, 0), make_float4(1, 0, 0, 0), make_float4(0, 1, 0, 0), iters, out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best_t) best_t = ms;
        }
        const double slabs = (double)blocks * 256 * iters * 8, tris = (double)blocks * 256 * iters * 2;
        printf("{\"cus\": %d, \"waves_per_simd\": %d, \"slab_tests_G_per_s\": %.1f, \"triangle_tests_G_per_s\": %.1f, \"slab_ms\": %.3f, \"tri_ms\": %.3f}\n",
               cus, wps, slabs / best_s * 1e-6, tris / best_t * 1e-6, best_s, best_t);
    }
    return 0;
}
