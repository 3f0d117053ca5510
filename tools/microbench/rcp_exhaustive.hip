// rcp_exhaustive.hip — for EVERY float32 bit pattern, compare candidate fast reciprocals against the
// compiler's IEEE-correct 1.0f/x (correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float rcp_a(float x) {  // rcp + 2 Newton steps
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
__device__ __forceinline__ float rcp_b(float x) {  // rcp + 1 Newton step
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_c(float x) {  // rcp + Newton + residual correction with the first estimate
    float r0 = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r0, 1.0f);
    float r1 = __builtin_fmaf(e, r0, r0);
    float e1 = __builtin_fmaf(-x, r1, 1.0f);
    return __builtin_fmaf(e1, r0, r1);
}

__global__ void check(unsigned long long* bad, uint32_t* firstBad, uint32_t expLo, uint32_t expHi) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const uint32_t bits = (uint32_t)b;
        const uint32_t ex = (bits >> 23) & 0xffu;
        if (ex < expLo || ex > expHi) continue;
        const float x = __builtin_bit_cast(float, bits);
        const uint32_t ref = __builtin_bit_cast(uint32_t, 1.0f / x);
        const float c[3] = {rcp_a(x), rcp_b(x), rcp_c(x)};
        for (int k = 0; k < 3; ++k) {
            if (__builtin_bit_cast(uint32_t, c[k]) != ref) {
                if (atomicAdd(&bad[k], 1ull) == 0) firstBad[k] = bits;
            }
        }
    }
}

int main() {
    unsigned long long* dBad; uint32_t* dFirst;
    hipMalloc(&dBad, 3 * 8); hipMalloc(&dFirst, 3 * 4);
    const uint32_t ranges[][2] = {{1, 254}, {2, 252}, {4, 250}, {8, 246}, {27, 227}};
    for (auto& rg : ranges) {
        hipMemset(dBad, 0, 24); hipMemset(dFirst, 0, 12);
        hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, dBad, dFirst, rg[0], rg[1]);
        hipDeviceSynchronize();
        unsigned long long bad[3]; uint32_t first[3];
        hipMemcpy(bad, dBad, 24, hipMemcpyDeviceToHost); hipMemcpy(first, dFirst, 12, hipMemcpyDeviceToHost);
        printf("biased exponent in [%u,%u]: mismatches  2-step %llu (first 0x%08x)  1-step %llu (0x%08x)  residual %llu (0x%08x)\n",
               rg[0], rg[1], bad[0], first[0], bad[1], first[1], bad[2], first[2]);
    }
    return 0;
}
