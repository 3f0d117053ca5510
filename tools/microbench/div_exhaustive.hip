// div_exhaustive.hip — all 2^23 x 2^23 mantissa pairs (a, b in [1, 2)): candidate fast divisions vs IEEE a / b.
// Progress line every 2^18 b-mantissas. usage: div_exhaustive [b_chunks_to_run (of 32)] [first_chunk]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float rcp1(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

__global__ void check(unsigned long long* bad, uint32_t* firstBad, uint32_t bLo, uint32_t bCount) {
    // grid-stride over (b index within chunk) x (a mantissa): each thread takes one b and loops a stripe of a
    const uint32_t bi = blockIdx.x;                   // one b mantissa per block
    if (bi >= bCount) return;
    const uint32_t bm = bLo + bi;
    const float b = __builtin_bit_cast(float, 0x3f800000u | bm);
    const float r = rcp1(b);
    unsigned long long bad1 = 0, bad2 = 0;
    uint32_t f1 = 0, f2 = 0;
    for (uint32_t am = threadIdx.x; am < (1u << 23); am += blockDim.x) {
        const float a = __builtin_bit_cast(float, 0x3f800000u | am);
        const uint32_t ref = __builtin_bit_cast(uint32_t, a / b);
        const float q0 = a * r;
        const float rem = __builtin_fmaf(-b, q0, a);
        const float q1 = __builtin_fmaf(rem, r, q0);
        const float rem2 = __builtin_fmaf(-b, q1, a);
        const float q2 = __builtin_fmaf(rem2, r, q1);
        if (__builtin_bit_cast(uint32_t, q1) != ref) { if (!bad1) f1 = am; ++bad1; }
        if (__builtin_bit_cast(uint32_t, q2) != ref) { if (!bad2) f2 = am; ++bad2; }
    }
    if (bad1) { if (atomicAdd(&bad[0], bad1) == 0) { firstBad[0] = bm; firstBad[1] = f1; } }
    if (bad2) { if (atomicAdd(&bad[1], bad2) == 0) { firstBad[2] = bm; firstBad[3] = f2; } }
}

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 32;
    const int first = argc > 2 ? atoi(argv[2]) : 0;
    unsigned long long* dBad; uint32_t* dFirst;
    hipMalloc(&dBad, 16); hipMalloc(&dFirst, 16);
    hipMemset(dBad, 0, 16); hipMemset(dFirst, 0, 16);
    const uint32_t per = (1u << 23) / 32;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int c = first; c < first + chunks && c < 32; ++c) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(check, dim3(per), dim3(256), 0, 0, dBad, dFirst, (uint32_t)c * per, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long bad[2]; uint32_t fb[4];
        hipMemcpy(bad, dBad, 16, hipMemcpyDeviceToHost); hipMemcpy(fb, dFirst, 16, hipMemcpyDeviceToHost);
        printf("chunk %2d/32 (%.1f s): mismatches so far  1-correction %llu (b=0x%06x a=0x%06x)  2-corrections %llu (b=0x%06x a=0x%06x)\n", c, ms / 1e3,
               bad[0], fb[0], fb[1], bad[1], fb[2], fb[3]);
        fflush(stdout);
    }
    return 0;
}
