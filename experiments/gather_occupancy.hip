// Micro-benchmark: dependent random 64-byte gathers (one per lane per step, like a BVH child-pair fetch) as a function of
// resident waves per SIMD and of the table size.  Answers: is the trace kernels' rate limited by occupancy (latency) or by the
// memory pipeline's gather rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void chase(const float4 * tab, const int * idx, int iters, unsigned mask, float * out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid];
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        const float4 * p = tab + 4ull * (r & mask);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(a.w);
    }
    out[gid] = acc;
}

int main() {
    const int max_rec = 1 << 22;            // 4M records x 64 B = 256 MB
    const int threads = 256, iters = 200;
    std::vector<float> h((size_t)max_rec * 16); for (auto & v : h) v = (float)rand() / RAND_MAX;
    const int max_n = 256 * 10 * threads;
    std::vector<int> hi(max_n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, max_n * 4)); CHECK(hipMalloc(&out, max_n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), max_n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int sizes_log2[] = { 14, 16, 19, 22 };   // 1 MB, 4 MB, 32 MB, 256 MB tables
    for (int sl : sizes_log2) {
        for (int wps = 1; wps <= 8; wps++) {       // blocks per CU = waves per SIMD (4 waves per block, 4 SIMDs per CU)
            const int blocks = 256 * wps, n = blocks * threads;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, (unsigned)((1u << sl) - 1), out);
                hipEventRecord(b); CHECK(hipEventSynchronize(b));
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            double recs = (double)n * iters;
            printf("table %4d MB  waves/SIMD %d: %.3f ms  %.1f Grecords/s  %.2f TB/s  (%.0f ns per dependent step)\n", (64 << sl) >> 20, wps, best, recs / best * 1e-6, recs * 64 / best * 1e-9, best * 1e6 / iters);
        }
    }
    return 0;
}
