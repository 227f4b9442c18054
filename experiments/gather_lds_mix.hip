// Micro-benchmark: a dependent chase where a fraction of the steps is served from a 32 KB table in LDS (the top of a BVH) and the
// rest from a 32 MB table in global memory (vector L1 / L2).  How much does moving the hot records to LDS relieve the CU's
// divergent-gather pipeline?  Records are 64 B; LDS records are padded to 80 B so that random slots spread over the banks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define LDS_RECORDS 400      // x 80 B = 32 000 B
template <int PAD>   // record stride in float4 units: 4 = 64 B (no padding), 5 = 80 B
__global__ __launch_bounds__(256) void chase(const float4 * tab, const int * idx, int iters, unsigned mask, unsigned lds_threshold, float * out) {
    __shared__ float4 top[LDS_RECORDS * PAD];
    for (int i = threadIdx.x; i < LDS_RECORDS * PAD; i += blockDim.x) top[i] = tab[i];
    __syncthreads();
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = idx[gid];
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        float4 a, b, c, d;
        if ((r >> 8 & 1023u) < lds_threshold) {          // per-lane choice: divergent, like lanes at different tree depths
            const float4 * p = top + (size_t)PAD * ((r >> 4) % LDS_RECORDS);
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else {
            const float4 * p = tab + 4ull * (r & mask);
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        }
        acc += a.x + b.y + c.z + d.w;
        r = r * 1664525u + 1013904223u + (unsigned)__float_as_int(a.w);
    }
    out[gid] = acc;
}

int main() {
    const int n_rec = 1 << 19, threads = 256, iters = 200, blocks = 256 * 4, n = blocks * threads;
    std::vector<float> h((size_t)n_rec * 16); for (auto & v : h) v = (float)rand() / RAND_MAX;
    std::vector<int> hi(n); for (auto & v : hi) v = rand();
    float4 * tab; int * idx; float * out;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int pad = 4; pad <= 5; pad++) for (int pct : { 0, 25, 35, 50, 75, 100 }) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a));
            if (pad == 4) hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, (unsigned)(n_rec - 1), (unsigned)(pct * 1024 / 100), out);
            else          hipLaunchKernelGGL(chase<5>, dim3(blocks), dim3(threads), 0, 0, tab, idx, iters, (unsigned)(n_rec - 1), (unsigned)(pct * 1024 / 100), out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("LDS stride %d B, %3d %% of steps from LDS: %.3f ms  %.1f Gsteps/s\n", pad * 16, pct, best, (double)n * iters / best * 1e-6);
    }
    return 0;
}
