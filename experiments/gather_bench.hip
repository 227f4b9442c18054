// Micro-benchmark: how fast can a wave gather one random 64-byte record per lane (a BVH child pair)?
//  A: each lane issues 4 x dwordx4 to its own record (64 distinct lines per instruction)
//  B: quad-cooperative: in instruction j the 4 lanes of a quad read the 4 x 16 B pieces of quad-mate j's record
//     (16 distinct 64-byte segments per instruction), pieces exchanged through LDS
//  C: like B but with global_load_lds (direct to LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void kA(const float4 * tab, const int * idx, int iters, int n_rec, float * out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid];
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        const float4 * p = tab + 4ull * (r % n_rec);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(a.w);   // dependent chain like a traversal
    }
    out[gid] = acc;
}
__global__ __launch_bounds__(256) void kB(const float4 * tab, const int * idx, int iters, int n_rec, float * out) {
    __shared__ float4 lds[4][4][64];   // [wave][j][lane]
    int gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned r = idx[gid];
    float acc = 0;
    const int qi = lane & 3, qb = lane & ~3;
    for (int it = 0; it < iters; it++) {
        unsigned rec = r % n_rec;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned rj = __shfl(rec, qb + j);
            float4 v = tab[4ull * rj + qi];
            lds[wave][j][lane] = v;            // piece qi of mate j's record
        }
        // my record's piece k sits at lds[wave][qi][qb + k]
        float4 a = lds[wave][qi][qb + 0], b = lds[wave][qi][qb + 1], c = lds[wave][qi][qb + 2], d = lds[wave][qi][qb + 3];
        acc += a.x + b.y + c.z + d.w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(a.w);
    }
    out[gid] = acc;
}
__global__ __launch_bounds__(256) void kC(const float4 * tab, const int * idx, int iters, int n_rec, float * out) {
    __shared__ float4 lds[4][4][64];
    int gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned r = idx[gid];
    float acc = 0;
    const int qi = lane & 3, qb = lane & ~3;
    for (int it = 0; it < iters; it++) {
        unsigned rec = r % n_rec;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned rj = __shfl(rec, qb + j);
            __builtin_amdgcn_global_load_lds((const void *)(tab + 4ull * rj + qi), (__attribute__((address_space(3))) void *)&lds[wave][j][0], 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.
        float4 a = lds[wave][qi][qb + 0], b = lds[wave][qi][qb + 1], c = lds[wave][qi][qb + 2], d = lds[wave][qi][qb + 3];
        acc += a.x + b.y + c.z + d.w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(a.w);
    }
    out[gid] = acc;
}

int main() {
    const int n_rec = 131072 * 4;           // 512k records x 64 B = 32 MB table (like the cfg3 BVH)
    const int blocks = 256 * 5, threads = 256, n = blocks * threads, iters = 200;
    std::vector<float> h((size_t)n_rec * 16); for (auto & v : h) v = (float)rand() / RAND_MAX;
    std::vector<int> hi(n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int k = 0; k < 3; k++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (k == 0) hipLaunchKernelGGL(kA, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, n_rec, out);
            if (k == 1) hipLaunchKernelGGL(kB, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, n_rec, out);
            if (k == 2) hipLaunchKernelGGL(kC, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, n_rec, out);
            hipEventRecord(b); CHECK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            double recs = (double)n * iters;
            printf("%c rep%d: %.3f ms  %.2f Grecords/s  %.2f TB/s\n", 'A' + k, rep, ms, recs / ms * 1e-6, recs * 64 / ms * 1e-9);
        }
    }
    std::vector<float> ho(n); hipMemcpy(ho.data(), out, n * 4, hipMemcpyDeviceToHost);
    double s = 0; for (float v : ho) s += v; printf("checksum %.3f\n", s);
    return 0;
}
