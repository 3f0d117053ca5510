// math_exhaustive.hip — proof by exhaustion, on the GPU it runs on, that the device fast paths of
// ptm::rcp and ptm::sqrt (csrc/ptmath.h) return the IEEE-correct result for EVERY float32 bit pattern:
// compared against hipcc's correctly rounded 1.0f/x and sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt),
// which are also what the CPU oracle computes. NaN results must be NaN on both sides (payload ignored).
// Prints one line: "rcp_mismatch=<n> sqrt_mismatch=<n> checked=<patterns>".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ptmath.h"

__device__ __forceinline__ bool same(float a, float b) {
    if (a != a && b != b) return true;
    return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b);
}

__global__ void check(unsigned long long* bad, uint32_t* firstBad) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        if (!same(ptm::rcp(x), 1.0f / x))
            if (atomicAdd(&bad[0], 1ull) == 0) firstBad[0] = (uint32_t)b;
        if (!same(ptm::sqrt(x), __builtin_sqrtf(x)))
            if (atomicAdd(&bad[1], 1ull) == 0) firstBad[1] = (uint32_t)b;
    }
}

int main() {
    unsigned long long* dBad;
    uint32_t* dFirst;
    if (hipMalloc(&dBad, 16) != hipSuccess || hipMalloc(&dFirst, 8) != hipSuccess) {
        fprintf(stderr, "no device\n");
        return 2;
    }
    hipMemset(dBad, 0, 16);
    hipMemset(dFirst, 0, 8);
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, dBad, dFirst);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    unsigned long long bad[2];
    uint32_t first[2];
    hipMemcpy(bad, dBad, 16, hipMemcpyDeviceToHost);
    hipMemcpy(first, dFirst, 8, hipMemcpyDeviceToHost);
    printf("rcp_mismatch=%llu sqrt_mismatch=%llu checked=4294967296 first_rcp=0x%08x first_sqrt=0x%08x\n", bad[0], bad[1], first[0],
           first[1]);
    return (bad[0] || bad[1]) ? 1 : 0;
}
