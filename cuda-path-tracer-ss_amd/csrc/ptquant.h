// ptquant.h — the 8-bit tone-mapped sample of writeToPixelsKernel (CudaTracer/CudaTracer.cu:72-85), literal and fast.
//
//   literal:  v = clamp(radiance, 0, 1);  v = pow(v, 1/2.2);  v = clamp(255 v + 0.5, 0, 255);  (uint)v     (NaN -> 0)
//
// The result has 256 possible values and the function is a non-decreasing step function of the radiance, so it is
// fully described by 255 thresholds T[k] = the smallest float32 whose sample is >= k. The device fast path computes a
// first guess k0 with the hardware log2/exp2 approximations (5 instructions instead of the ~90 of the exact software
// pow), which is never off by more than one step, and settles it with two exact comparisons against T[k0] and
// T[k0 + 1]. Equality with the literal function is established for EVERY float32 bit pattern by exhaustion on the
// GPU the tests run on (tests/csrc/math_exhaustive.hip, tests/test_gpu_math.py); the table itself is built on the host
// from the literal function (same ptm::pow, bit-identical on both sides), at ptss_create.
#ifndef PTSS_PTQUANT_H
#define PTSS_PTQUANT_H
#include "ptmath.h"

namespace ptq {

constexpr int kTableFloats = 260;  // T[0] = -inf, T[1..255], T[256] = NaN (unreachable), padded to whole float4 rows (65)

PTM_HD uint32_t quantize_literal(float radiance) {
    float v = ptm::clamp(radiance, 0.0f, 1.0f);
    v = ptm::pow(v, ptm::kGamma);
    v = ptm::clamp(255 * v + 0.5f, 0.f, 255.f);
    return (v == v) ? (uint32_t)v : 0u;  // NaN -> 0 (CUDA's float -> uint of NaN)
}

// Host: T[k] by bisection over the bit patterns of [0, 1] (non-negative floats order like their bit patterns).
// Returns false if the literal function is not non-decreasing at the probed points (it is; the exhaustive device
// test would also catch any disagreement).
inline bool build_thresholds(float* T) {
    T[0] = -ptm::inf();
    for (int k = 1; k <= 255; ++k) {
        uint32_t lo = 0u, hi = 0x3f800000u;  // literal(0) = 0 < k <= 255 = literal(1)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (quantize_literal(ptm::u2f(mid)) >= (uint32_t)k) hi = mid; else lo = mid;
        }
        T[k] = ptm::u2f(hi);
        if (k > 1 && !(T[k] >= T[k - 1])) return false;
    }
    for (int k = 256; k < kTableFloats; ++k) T[k] = ptm::qnan();  // nothing compares >= NaN: +inf must stay at 255
    return true;
}

PTM_HD uint32_t quantize_fast(float radiance, const float* T) {
#if defined(__HIP_DEVICE_COMPILE__)
    // first guess: 255 * 2^(log2(x) / 2.2) + 0.5 with v_log_f32 / v_exp_f32; NaN, negatives and zero come out as 0,
    // anything >= 1 as 255 (v_med3_f32 returns the smaller bound for a NaN)
    const float guess = __builtin_fmaf(255.0f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(radiance) * ptm::kGamma), 0.5f);
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_fmed3f(guess, 0.0f, 255.0f);
    uint32_t k = k0;
    k -= (radiance < T[k0]) ? 1u : 0u;        // T[0] = -inf: never below it
    k += (radiance >= T[k0 + 1]) ? 1u : 0u;   // T[256] = NaN: never reached, not even by +inf
    return k;
#else
    (void)T;
    return quantize_literal(radiance);  // the host has no use for the table form
#endif
}

}  // namespace ptq
#endif
