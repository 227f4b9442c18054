// Micro-benchmark: a WAVE-UNIFORM dependent chase over 64-byte records (what a wave-packet BVH walk does: every lane of the wave
// needs the same child pair) through four fetch mechanisms:
//   S  scalar load (s_load_dwordx16 through the scalar data cache) into SGPRs
//   V  vector load, every lane the same address (global_load_dwordx4 x 4): TA sees one line, returns 64 copies
//   Q  4 lanes load one dwordx4 each, 16 v_readlane broadcast the record into SGPRs
//   L  4 lanes load one dwordx4 each, write 64 B to LDS, every lane reads them back (broadcast ds_read_b128 x 4)
// as a function of table size and resident waves per SIMD.  Each step also runs `work` dependent FMAs per lane on the record
// (stand-in for the two slab tests), so the rates show what the fetch path costs next to the arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v * cptr;
typedef const __attribute__((address_space(1))) f4v * gptr;

template <int MODE, int WORK>
__global__ __launch_bounds__(256) void chase(const float4 * tab, int iters, unsigned mask, float * out) {
    __shared__ f4v lds[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned r = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + wave) * 2654435761u);
    float acc = (float)lane;
    for (int it = 0; it < iters; it++) {
        const unsigned i = r & mask;
        f4v a, b, c, d;
        if (MODE == 0) { cptr p = (cptr)(const void *)tab + 4ull * i; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        else if (MODE == 1) { gptr p = (gptr)(const void *)tab + 4ull * i; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        else if (MODE == 2) {
            gptr p = (gptr)(const void *)tab + 4ull * i;
            f4v v = p[lane & 3];
            #define RL(x, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l))
            a = f4v{ RL(v.x, 0), RL(v.y, 0), RL(v.z, 0), RL(v.w, 0) }; b = f4v{ RL(v.x, 1), RL(v.y, 1), RL(v.z, 1), RL(v.w, 1) };
            c = f4v{ RL(v.x, 2), RL(v.y, 2), RL(v.z, 2), RL(v.w, 2) }; d = f4v{ RL(v.x, 3), RL(v.y, 3), RL(v.z, 3), RL(v.w, 3) };
        } else {
            gptr p = (gptr)(const void *)tab + 4ull * i;
            if (lane < 4) lds[wave][lane] = p[lane];
            __builtin_amdgcn_wave_barrier();
            a = lds[wave][0]; b = lds[wave][1]; c = lds[wave][2]; d = lds[wave][3];
            __builtin_amdgcn_wave_barrier();
        }
        float x = acc;
        #pragma unroll
        for (int w = 0; w < WORK; w += 8) {
            x = x * a.x + a.y; x = x * b.x + b.y; x = x * c.x + c.y; x = x * d.x + d.y;
            x = x * a.z + b.z; x = x * c.z + d.z; x = x * a.y + c.y; x = x * b.y + d.y;
        }
        acc = x;
        r = r * 1664525u + 1013904223u + (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.w));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int WORK>
static void run(const char * name, const float4 * tab, float * out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 400;
    const int sizes_log2[] = { 7, 11, 14, 19 };          // 8 KB, 128 KB, 1 MB, 32 MB
    for (int sl : sizes_log2) for (int wps : { 1, 2, 4, 8 }) {
        const int blocks = 256 * wps;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL((chase<MODE, WORK>), dim3(blocks), dim3(256), 0, 0, tab, iters, (unsigned)((1u << sl) - 1), out);
            hipEventRecord(b); CHECK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double steps = (double)blocks * 4 * iters;
        printf("%s work %2d  table %6d KB  waves/SIMD %d: %.3f ms  %.2f G wave-steps/s  (%.0f ns per dependent step)\n", name, WORK, (64 << sl) >> 10, wps, best,
               steps / best * 1e-6, best * 1e6 / iters);
    }
}

int main() {
    const int max_rec = 1 << 19;
    std::vector<float> h((size_t)max_rec * 16); for (auto & v : h) v = (float)rand() / RAND_MAX;
    float4 * tab; float * out;
    CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0, 0>("S scalar     ", tab, out); run<1, 0>("V vec-uniform", tab, out); run<2, 0>("Q quad+rdlane", tab, out); run<3, 0>("L quad+lds   ", tab, out);
    run<0, 40>("S scalar     ", tab, out); run<1, 40>("V vec-uniform", tab, out); run<2, 40>("Q quad+rdlane", tab, out); run<3, 40>("L quad+lds   ", tab, out);
    return 0;
}
