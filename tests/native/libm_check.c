/* Exhaustive / randomised bit-for-bit comparison of csrc/rtx_libm.h against the host libm.
 * usage: libm_check acosf|atanf|expf|log2f [stride]   sweeps ALL 2^32 bit patterns with the given stride (default 1 = exhaustive)
 *        libm_check atan2f [count]        count pseudo-random argument pairs of four kinds
 * prints "<name> checked=N mismatches=M" */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "../../cpu-raytracer_amd/csrc/rtx_libm.h"

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

typedef struct { int mode; uint64_t lo, hi, stride; uint64_t mism; uint32_t first_bad[2]; } job_t;

static void * run(void * p) {
    job_t * j = (job_t *)p; j->mism = 0;
    if (j->mode == 0) { for (uint64_t u = j->lo; u < j->hi; u += j->stride) { float x = fb((uint32_t)u); if (!same(rtx_acosf(x), acosf(x))) { if (!j->mism) j->first_bad[0] = (uint32_t)u; j->mism++; } } }
    if (j->mode == 1) { for (uint64_t u = j->lo; u < j->hi; u += j->stride) { float x = fb((uint32_t)u); if (!same(rtx_atanf(x), atanf(x))) { if (!j->mism) j->first_bad[0] = (uint32_t)u; j->mism++; } } }
    if (j->mode == 3) { for (uint64_t u = j->lo; u < j->hi; u += j->stride) { float x = fb((uint32_t)u); if (!same(rtx_expf(x), expf(x))) { if (!j->mism) j->first_bad[0] = (uint32_t)u; j->mism++; } } }
    if (j->mode == 4) { for (uint64_t u = j->lo; u < j->hi; u += j->stride) { float x = fb((uint32_t)u); if (!same(rtx_log2f(x), log2f(x))) { if (!j->mism) j->first_bad[0] = (uint32_t)u; j->mism++; } } }
    if (j->mode == 2) {
        uint64_t s = 0x9E3779B97F4A7C15ull * (j->lo + 1);
        for (uint64_t i = j->lo; i < j->hi; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            uint32_t a = (uint32_t)s, b = (uint32_t)(s >> 32);
            float y, x;
            switch (i & 3) {
                case 0: y = fb(a); x = fb(b); break;                                        /* any bit patterns */
                case 1: y = ((int32_t)a) * (1.0f / 2147483648.0f); x = ((int32_t)b) * (1.0f / 2147483648.0f); break;   /* unit square */
                case 2: y = fb((a & 0x807fffffu) | 0x3f000000u); x = fb((b & 0x807fffffu) | 0x3f000000u); break;       /* same exponent */
                default: y = ((int32_t)a) * (1.0f / 2147483648.0f); x = fb((b & 0x80000000u) | 0x3f800000u); break;    /* x = +-1 */
            }
            if (!same(rtx_atan2f(y, x), atan2f(y, x))) { if (!j->mism) { j->first_bad[0] = bits(y); j->first_bad[1] = bits(x); } j->mism++; }
        }
    }
    return NULL;
}

int main(int argc, char ** argv) {
    if (argc < 2) return 2;
    int mode = !strcmp(argv[1], "acosf") ? 0 : !strcmp(argv[1], "atanf") ? 1 : !strcmp(argv[1], "expf") ? 3 : !strcmp(argv[1], "log2f") ? 4 : 2;
    uint64_t total = mode == 2 ? (argc > 2 ? strtoull(argv[2], 0, 0) : 400000000ull) : (1ull << 32);
    uint64_t stride = (mode != 2 && argc > 2) ? strtoull(argv[2], 0, 0) : 1;
    if (stride < 1) stride = 1;
    int nt = 8; pthread_t th[8]; job_t jobs[8]; uint64_t mism = 0;
    for (int t = 0; t < nt; t++) { jobs[t].mode = mode; jobs[t].stride = stride; jobs[t].lo = total * t / nt; jobs[t].hi = total * (t + 1) / nt; pthread_create(&th[t], 0, run, &jobs[t]); }
    for (int t = 0; t < nt; t++) { pthread_join(th[t], 0); if (jobs[t].mism && !mism) printf("first mismatch: %08x %08x\n", jobs[t].first_bad[0], jobs[t].first_bad[1]); mism += jobs[t].mism; }
    printf("%s checked=%llu mismatches=%llu\n", argv[1], (unsigned long long)(total / stride), (unsigned long long)mism);
    return mism ? 1 : 0;
}
