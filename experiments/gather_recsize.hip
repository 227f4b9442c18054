// Micro-benchmark: dependent random gathers of 16 / 32 / 64 / 128-byte records (1 / 2 / 4 / 8 dwordx4 per lane per step) from a
// 32 MB table.  Is the chip's gather rate bound by records (cache lines touched) or by bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int Q>   // Q float4 per record
__global__ __launch_bounds__(256) void chase(const float4 * tab, const int * idx, int iters, unsigned mask, float * out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid];
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        const float4 * p = tab + (size_t)Q * (r & mask);
        float4 v[Q];
        #pragma unroll
        for (int k = 0; k < Q; k++) v[k] = p[k];
        #pragma unroll
        for (int k = 0; k < Q; k++) acc += v[k].x;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(v[0].w);
    }
    out[gid] = acc;
}

template <int Q> void run(const float4 * tab, const int * idx, float * out, size_t table_bytes) {
    const int threads = 256, iters = 200, blocks = 256 * 4, n = blocks * threads;
    unsigned recs = (unsigned)(table_bytes / (16 * Q));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(chase<Q>, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, recs - 1, out);
        hipEventRecord(b); CHECK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    double total = (double)n * iters;
    printf("record %3d B: %.3f ms  %.1f Grecords/s  %.2f TB/s\n", 16 * Q, best, total / best * 1e-6, total * 16 * Q / best * 1e-9);
}

int main() {
    const size_t bytes = 32u << 20;
    std::vector<float> h(bytes / 4); for (auto & v : h) v = (float)rand() / RAND_MAX;
    const int n = 256 * 4 * 256;
    std::vector<int> hi(n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out;
    CHECK(hipMalloc(&tab, bytes)); CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    run<1>(tab, idx, out, bytes); run<2>(tab, idx, out, bytes); run<4>(tab, idx, out, bytes); run<8>(tab, idx, out, bytes);
    return 0;
}
