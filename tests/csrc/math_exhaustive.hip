// math_exhaustive.hip — proof by exhaustion, on the GPU it runs on, that the device fast paths of ptm::rcp,
// ptm::sqrt and ptm::div (csrc/ptmath.h) return the IEEE-correct result:
//   rcp, sqrt : every one of the 2^32 float32 bit patterns, against hipcc's correctly rounded 1.0f/x and sqrtf
//               (-fhip-fp32-correctly-rounded-divide-sqrt) — which are also what the CPU oracle computes;
//   div       : every one of the 2^23 x 2^23 mantissa pairs with a, b in [1, 2) (a quotient's bits depend only on the
//               mantissas as long as nothing under/overflows), then the full guarded function (incl. its IEEE escape)
//               on every b-mantissa x 4096 a-mantissas for a grid of exponent pairs that straddles the guard range,
//               plus zeros / infinities / NaNs. A control column counts how often the UNcorrected quotient a*r differs,
//               to show the comparison is not vacuous.
//   quantize  : the table form of the 8-bit tone map (csrc/ptquant.h) against the literal clamp/pow/scale sequence for every
//               one of the 2^32 bit patterns; a control counts how often the hardware first guess alone is off.
// NaN results must be NaN on both sides (payload ignored).
// usage: ptss_mathcheck [div_chunks (0..32, default 32)]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ptmath.h"
#include "ptquant.h"

__device__ __forceinline__ bool same(float a, float b) {
    if (a != a && b != b) return true;
    return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b);
}

__global__ void checkUnary(unsigned long long* bad, uint32_t* firstBad) {
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

__global__ void checkQuantize(unsigned long long* bad, const float* T) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const uint32_t want = ptq::quantize_literal(x);
        if (ptq::quantize_fast(x, T) != want) atomicAdd(&bad[6], 1ull);
        const float guess = __builtin_fmaf(255.0f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * ptm::kGamma), 0.5f);
        if ((uint32_t)__builtin_amdgcn_fmed3f(guess, 0.0f, 255.0f) != want) atomicAdd(&bad[7], 1ull);  // control
    }
}

// one b mantissa per block, all 2^23 a mantissas across its threads
__global__ void checkDivMantissas(unsigned long long* bad, uint32_t bLo, uint32_t bCount) {
    const uint32_t bi = blockIdx.x;
    if (bi >= bCount) return;
    const float b = __builtin_bit_cast(float, 0x3f800000u | (bLo + bi));
    unsigned long long wrong = 0, control = 0;
    const float r = ptm::rcp(b);
    for (uint32_t am = threadIdx.x; am < (1u << 23); am += blockDim.x) {
        const float a = __builtin_bit_cast(float, 0x3f800000u | am);
        const float ref = a / b;
        if (!same(ptm::div(a, b), ref)) ++wrong;
        if (!same(a * r, ref)) ++control;
    }
    if (wrong) atomicAdd(&bad[2], wrong);
    if (control) atomicAdd(&bad[3], control);
}

// the guarded function across exponents: block = (exponent pair, b mantissa chunk)
__global__ void checkDivExponents(unsigned long long* bad, const int* ea, const int* eb, int pairs) {
    const int pair = blockIdx.y;
    if (pair >= pairs) return;
    unsigned long long wrong = 0;
    for (uint32_t bm = blockIdx.x * blockDim.x + threadIdx.x; bm < (1u << 23); bm += gridDim.x * blockDim.x) {
        const float b = __builtin_bit_cast(float, ((uint32_t)(eb[pair] + 127) << 23) | bm);
        for (uint32_t k = 0; k < 64; ++k) {
            const uint32_t am = (bm * 2654435761u + k * 40503u * 2057u) & 0x7fffffu;
            float a = __builtin_bit_cast(float, ((uint32_t)(ea[pair] + 127) << 23) | am);
            if (k & 1) a = -a;
            if (!same(ptm::div(a, b), a / b)) ++wrong;
            float qx, qy, qz;
            ptm::div3(a, -a, a * 0.75f, b, qx, qy, qz);
            if (!same(qx, a / b) || !same(qy, -a / b) || !same(qz, (a * 0.75f) / b)) ++wrong;
        }
    }
    if (wrong) atomicAdd(&bad[4], wrong);
}

__global__ void checkDivSpecials(unsigned long long* bad) {
    const float v[] = {0.0f, -0.0f, 1.0f, -1.0f, 3.0f, 1e-45f, 1e-39f, 1.17549435e-38f, 3.4028235e38f, 1e30f, 1e-30f,
                       __builtin_huge_valf(), -__builtin_huge_valf(), __builtin_nanf(""), 8.6736174e-19f, 1.1529215e18f};
    const int n = sizeof(v) / sizeof(v[0]);
    const int i = threadIdx.x / n, j = threadIdx.x % n;
    if (i < n) {
        if (!same(ptm::div(v[i], v[j]), v[i] / v[j])) atomicAdd(&bad[5], 1ull);
        float qx, qy, qz;
        ptm::div3(v[i], v[j], 1.0f, v[j], qx, qy, qz);
        if (!same(qx, v[i] / v[j]) || !same(qy, v[j] / v[j]) || !same(qz, 1.0f / v[j])) atomicAdd(&bad[5], 1ull);
    }
}

int main(int argc, char** argv) {
    const int divChunks = argc > 1 ? atoi(argv[1]) : 32;
    unsigned long long* dBad;
    uint32_t* dFirst;
    if (hipMalloc(&dBad, 8 * 8) != hipSuccess || hipMalloc(&dFirst, 8) != hipSuccess) {
        fprintf(stderr, "no device\n");
        return 2;
    }
    hipMemset(dBad, 0, 64);
    hipMemset(dFirst, 0, 8);
    hipLaunchKernelGGL(checkUnary, dim3(256 * 16), dim3(256), 0, 0, dBad, dFirst);
    if (hipDeviceSynchronize() != hipSuccess) return 3;

    float hT[ptq::kTableFloats];
    const bool monotone = ptq::build_thresholds(hT);
    float* dT;
    hipMalloc(&dT, sizeof(hT));
    hipMemcpy(dT, hT, sizeof(hT), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(checkQuantize, dim3(256 * 16), dim3(256), 0, 0, dBad, dT);
    if (hipDeviceSynchronize() != hipSuccess) return 3;

    const uint32_t per = (1u << 23) / 32;
    for (int c = 0; c < divChunks && c < 32; ++c) {
        hipLaunchKernelGGL(checkDivMantissas, dim3(per), dim3(256), 0, 0, dBad, (uint32_t)c * per, per);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        fprintf(stderr, "div mantissa chunk %d/32 done\n", c + 1);  // progress (a full run takes ~45 s)
    }

    // exponent grid straddling the guard [2^-60, 2^60): inside, on the edges, outside, and quotients near the float range ends
    const int es[] = {-126, -100, -61, -60, -59, -1, 0, 1, 59, 60, 61, 100, 127};
    int ea[169], eb[169], pairs = 0;
    for (int x : es)
        for (int y : es) {
            ea[pairs] = x;
            eb[pairs] = y;
            ++pairs;
        }
    int *dEa, *dEb;
    hipMalloc(&dEa, sizeof(ea));
    hipMalloc(&dEb, sizeof(eb));
    hipMemcpy(dEa, ea, sizeof(ea), hipMemcpyHostToDevice);
    hipMemcpy(dEb, eb, sizeof(eb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(checkDivExponents, dim3(64, pairs), dim3(256), 0, 0, dBad, dEa, dEb, pairs);
    hipLaunchKernelGGL(checkDivSpecials, dim3(1), dim3(256), 0, 0, dBad);
    if (hipDeviceSynchronize() != hipSuccess) return 3;

    unsigned long long bad[8];
    uint32_t first[2];
    hipMemcpy(bad, dBad, 64, hipMemcpyDeviceToHost);
    hipMemcpy(first, dFirst, 8, hipMemcpyDeviceToHost);
    printf("rcp_mismatch=%llu sqrt_mismatch=%llu checked=4294967296 first_rcp=0x%08x first_sqrt=0x%08x\n", bad[0], bad[1], first[0],
           first[1]);
    printf("div_mantissa_mismatch=%llu div_pairs_checked=%llu control_uncorrected_mismatch=%llu div_exponent_mismatch=%llu "
           "div_special_mismatch=%llu\n",
           bad[2], (unsigned long long)(divChunks > 32 ? 32 : divChunks) * per * (1ull << 23), bad[3], bad[4], bad[5]);
    printf("quant_mismatch=%llu quant_checked=4294967296 quant_guess_alone_mismatch=%llu quant_table_monotone=%d\n", bad[6], bad[7],
           monotone ? 1 : 0);
    return (bad[0] || bad[1] || bad[2] || bad[4] || bad[5] || bad[6] || !monotone) ? 1 : 0;
}
