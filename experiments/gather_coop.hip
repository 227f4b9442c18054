// Micro-benchmark: 64 lanes each need their OWN random 128-byte record (a 4-wide BVH record, a pair of triangles).
//   direct: every lane issues 8 x global_load_dwordx4 on its record — every instruction touches 64 distinct lines;
//   coop:   instruction j loads records 8j .. 8j+7 of the wave, 8 lanes per record (one 16-byte chunk each: 8 distinct lines per instruction),
//           the chunks go through LDS (swizzled, conflict-free) and every lane reads its own record back.
// Same dependent chain as gather_occupancy.hip (the next address depends on the data).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void direct(const float4 * tab, const int * idx, int iters, unsigned mask, float * out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid];
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        const float4 * p = tab + 8ull * (r & mask);
        float4 v[8];
        for (int c = 0; c < 8; c++) v[c] = p[c];
        for (int c = 0; c < 8; c++) acc += v[c].x + v[c].y * 0.5f + v[c].z + v[c].w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(v[0].w) + (unsigned)__float_as_int(v[7].x);
    }
    out[gid] = acc;
}

template <int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void coop(const float4 * tab, const int * idx, int iters, unsigned mask, float * out) {
    __shared__ float4 xbuf[WAVES_PER_BLOCK][64 * 8];
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 * const xb = xbuf[wave];
    unsigned r = idx[gid];
    float acc = 0;
    const int grp = lane >> 3, chunk = lane & 7;
    for (int it = 0; it < iters; it++) {
        const unsigned rec = r & mask;
        float4 v[8];
        for (int j = 0; j < 8; j++) {
            const unsigned other = (unsigned)__builtin_amdgcn_ds_bpermute((8 * j + grp) << 2, (int)rec);      // record of lane 8j + grp
            v[j] = tab[8ull * other + chunk];
        }
        for (int j = 0; j < 8; j++) { const int rr = 8 * j + grp; xb[rr * 8 + (chunk ^ (rr & 7))] = v[j]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed (one wave per buffer: no barrier)
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < 8; c++) v[c] = xb[lane * 8 + (c ^ (lane & 7))];
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < 8; c++) acc += v[c].x + v[c].y * 0.5f + v[c].z + v[c].w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(v[0].w) + (unsigned)__float_as_int(v[7].x);
    }
    out[gid] = acc;
}

int main() {
    const int max_rec = 1 << 18;            // 256 k records x 128 B = 32 MB
    const int iters = 200;
    std::vector<float> h((size_t)max_rec * 32); for (auto & v : h) v = (float)(rand() & 0xffff) / 65536.0f;
    const int max_n = 256 * 16 * 64;
    std::vector<int> hi(max_n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out, * out2;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, max_n * 4)); CHECK(hipMalloc(&out, max_n * 4)); CHECK(hipMalloc(&out2, max_n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), max_n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int sl : { 13, 18 }) {              // 1 MB, 32 MB
        for (int wps = 1; wps <= 4; wps++) {   // waves per SIMD (one wave per block: 4 blocks per CU per wave/SIMD)
            const int blocks = 256 * 4 * wps, n = blocks * 64;
            float best[2] = { 1e30f, 1e30f };
            for (int rep = 0; rep < 3; rep++) for (int k = 0; k < 2; k++) {
                CHECK(hipEventRecord(a));
                if (k == 0) hipLaunchKernelGGL(direct<1>, dim3(blocks), dim3(64), 0, 0, tab, idx, iters, (unsigned)((1u << sl) - 1), out);
                else        hipLaunchKernelGGL(coop<1>,   dim3(blocks), dim3(64), 0, 0, tab, idx, iters, (unsigned)((1u << sl) - 1), out2);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best[k]) best[k] = ms;
            }
            std::vector<float> o1(n), o2(n);
            CHECK(hipMemcpy(o1.data(), out, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2.data(), out2, n * 4, hipMemcpyDeviceToHost));
            int bad = 0; for (int i = 0; i < n; i++) if (o1[i] != o2[i]) bad++;
            double recs = (double)n * iters;
            printf("table %3d MB  waves/SIMD %d:  direct %.3f ms %.1f G records/s (%.0f ns per step)   coop %.3f ms %.1f G records/s (%.0f ns per step)   mismatches %d\n",
                   (128 << sl) >> 20, wps, best[0], recs / best[0] * 1e-6, best[0] * 1e6 / iters, best[1], recs / best[1] * 1e-6, best[1] * 1e6 / iters, bad);
        }
    }
    return 0;
}
