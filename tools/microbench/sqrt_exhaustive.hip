// sqrt_exhaustive.hip — every non-negative float32: candidate fast square roots vs the compiler's IEEE sqrtf.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float sq_a(float x) {  // rsq + one Goldschmidt/Newton correction
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    const float h = 0.5f * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sq_b(float x) {  // hardware sqrt + correction with rsq-based half reciprocal
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sq_c(float x) {  // rsq, two corrections
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    s = __builtin_fmaf(s, e, s);
    h = __builtin_fmaf(h, e, h);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}

__global__ void check(unsigned long long* bad, uint32_t* firstBad, uint32_t expLo, uint32_t expHi) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 31); b += stride) {
        const uint32_t bits = (uint32_t)b;
        const uint32_t ex = (bits >> 23) & 0xffu;
        if (ex < expLo || ex > expHi) continue;
        const float x = __builtin_bit_cast(float, bits);
        const uint32_t ref = __builtin_bit_cast(uint32_t, __builtin_sqrtf(x));
        const float c[3] = {sq_a(x), sq_b(x), sq_c(x)};
        for (int k = 0; k < 3; ++k)
            if (__builtin_bit_cast(uint32_t, c[k]) != ref)
                if (atomicAdd(&bad[k], 1ull) == 0) firstBad[k] = bits;
    }
}

int main() {
    unsigned long long* dBad; uint32_t* dFirst;
    hipMalloc(&dBad, 3 * 8); hipMalloc(&dFirst, 3 * 4);
    const uint32_t ranges[][2] = {{1, 254}, {2, 253}, {16, 240}, {32, 222}};
    for (auto& rg : ranges) {
        hipMemset(dBad, 0, 24); hipMemset(dFirst, 0, 12);
        hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, dBad, dFirst, rg[0], rg[1]);
        hipDeviceSynchronize();
        unsigned long long bad[3]; uint32_t first[3];
        hipMemcpy(bad, dBad, 24, hipMemcpyDeviceToHost); hipMemcpy(first, dFirst, 12, hipMemcpyDeviceToHost);
        printf("biased exponent in [%u,%u]: mismatches  rsq+1 %llu (first 0x%08x)  sqrt+rsq %llu (0x%08x)  rsq+2 %llu (0x%08x)\n",
               rg[0], rg[1], bad[0], first[0], bad[1], first[1], bad[2], first[2]);
    }
    return 0;
}
