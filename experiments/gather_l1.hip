// Micro-benchmark: what is a gather worth when it HITS — the vector L1 of its CU (tables of 16 ... 256 KB), or the same line as its
// neighbours' (64 / 16 / 4 / 1 distinct 64-byte records per wave and step)?  Same dependent chain as gather_occupancy.hip, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void chase(const float4 * tab, const int * idx, int iters, unsigned mask, int share_shift, float * out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid >> share_shift];          // lanes of a group of 2^share_shift start (and stay) on the same chain
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
    const int max_rec = 1 << 19;            // 32 MB
    const int threads = 256, iters = 200;
    std::vector<float> h((size_t)max_rec * 16); for (auto & v : h) v = (float)rand() / RAND_MAX;
    const int blocks = 256 * 4, n = blocks * threads;
    std::vector<int> hi(n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int sizes_log2[] = { 8, 9, 10, 12, 14, 19 };   // 16 KB, 32 KB, 64 KB, 256 KB, 1 MB, 32 MB
    for (int sl : sizes_log2) {
        for (int share = 0; share <= 6; share += 2) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, (unsigned)((1u << sl) - 1), share, out);
                hipEventRecord(b); CHECK(hipEventSynchronize(b));
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            double recs = (double)n * iters;
            printf("table %6d KB  distinct records per wave %2d: %.3f ms  %.1f G lane-records/s  (%.0f ns per dependent step)\n", (64 << sl) >> 10, 64 >> share, best, recs / best * 1e-6, best * 1e6 / iters);
        }
    }
    return 0;
}
