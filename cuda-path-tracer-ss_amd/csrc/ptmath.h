// ptmath.h — deterministic IEEE-f32 primitives shared by the gfx950 kernels and the CPU oracle.
//
// Why this file exists (DESIGN.md "Parity"): one sample whose branch flips
// (`r < 0`, `discriminent < 0`, `dist <= 0`) moves a pixel by more than the 1e-4 budget, and
// libm (x86) and ROCm device-libs transcendentals differ in the last ulp. Everything here is
// built only from + - * / sqrt fma, float<->int conversion and bit casts — all correctly
// rounded on both x86-64 (SSE/FMA3) and gfx950 — so the same inputs give the same BITS on the
// host and on the GPU. Both sides must be compiled with -ffp-contract=off (fma only where it is
// written), -fno-fast-math, and (hipcc) -fhip-fp32-correctly-rounded-divide-sqrt.
//
// What replaces what in the reference (glm 0.9.5.4 / CUDA libm call sites):
//   dot/cross/normalize/quat*vec3/quat normalize  ... glm, used all over CudaTracer/CudaTracer.cu
//   ptm::sincos   ... cos/sin/cosf/sinf   CudaTracer/CudaTracer.cu:541,555,567-570
//   ptm::atan     ... atan                CudaTracer/CudaTracer.cu:564
//   ptm::log      ... log                 CudaTracer/CudaTracer.cu:564
//   ptm::exp      ... expf                CudaTracer/CudaTracer.cu:182-184
//   ptm::pow      ... pow                 CudaTracer/CudaTracer.cu:75-77,552
//   ptm::tan      ... tan                 CudaTracer/CudaTracer.cu:334
// Polynomial coefficients are the published Cephes single-precision minimax sets; accuracy
// (<= 2 ulp on the ranges the tracer uses, pinned against libm in tests/test_ptmath.py) is the
// same class as the CUDA libm the reference was built with.
//
// NaN policy: every select is written as an explicit ordered comparison, there is no fmin/fmax,
// and no float->int conversion is ever applied to a NaN/inf, so NaNs take the same path on both
// sides (payload bits may differ; tests compare NaN==NaN).
#pragma once
#include <stdint.h>
#include "ptss_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PTM_HD __host__ __device__ __forceinline__
#else
#define PTM_HD inline __attribute__((always_inline))
#endif

namespace ptm {

constexpr float kPi = 3.14159265358979323846f;        // RenderStructs.h:9 (float literal M_PI)
constexpr float kInvPi = 0.31830988618f;              // CudaTracer.h:4 INVERSE_PI
constexpr float kRayBump = 1e-4f;                     // CudaTracer.h:6 RAY_BUMP_EPSILON
constexpr float kGamma = (1 / 2.2f);                  // CudaTracer.h:7 GAMMA_CORRECTION

PTM_HD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PTM_HD float abs(float x) { return __builtin_fabsf(x); }

// sqrt(x) and rcp(x) = 1.0f / x, correctly rounded (IEEE) on both sides.
// Host: the plain operations. gfx950: hipcc's IEEE sequences cost ~38 / ~45 SIMD cycles each
// (v_div_scale / v_div_fmas / v_div_fixup are quarter-rate), and the tracer does ~36 reciprocals and
// ~18 square roots per ray-bounce. The device paths below are the hardware approximation plus ONE fma
// correction — 3 / 5 instructions — on the operand ranges where that is PROVEN bit-identical to the
// IEEE result by exhaustion over every float32 in the range (tests/csrc/math_exhaustive.hip, run by
// tests/test_gpu_math.py on the GPU the tests run on); outside the range they fall back to the IEEE
// sequence. Ranges (biased exponent): rcp [2, 252], sqrt [32, 222].
#if defined(__HIP_DEVICE_COMPILE__)
// Range guards are evaluated as wave masks: one ballot per DIRECT compare, combined with scalar ANDs and held against the
// mask of active lanes. (A compound bool handed to a ballot makes hipcc materialise it in a VGPR and compare it again:
// v_cndmask + v_cmp at every guarded operation.) `allLanes(m)`: every active lane has its bit set in m.
PTM_HD unsigned long long lanes(bool directCompare) { return __builtin_amdgcn_ballot_w64(directCompare); }
PTM_HD bool allLanes(unsigned long long m) { return m == __builtin_amdgcn_ballot_w64(true); }
#endif
PTM_HD float sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    const float h = 0.5f * y;
    const float r = __builtin_fmaf(-s, s, x);
    float out = __builtin_fmaf(r, h, s);
    const bool lo = x >= 2.5243549e-29f /* 2^-95 */, hi = x < 7.9228163e28f /* 2^96 */;
    // wave-uniform escape: only a wave holding an out-of-range operand runs the IEEE sequence at all
    if (__builtin_expect(!allLanes(lanes(lo) & lanes(hi)), 0)) out = (lo && hi) ? out : __builtin_sqrtf(x);
    return out;
#else
    return __builtin_sqrtf(x);
#endif
}
PTM_HD float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    float out = __builtin_fmaf(e, r0, r0);
    const float ax = __builtin_fabsf(x);
    const bool lo = ax >= 2.3509887e-38f /* 2^-125 */, hi = ax < 8.5070592e37f /* 2^126 */;
    if (__builtin_expect(!allLanes(lanes(lo) & lanes(hi)), 0)) out = (lo && hi) ? out : 1.0f / x;
    return out;
#else
    return 1.0f / x;
#endif
}
// rcp for a caller that DISCARDS the result whenever |x| <= 1e-7 (Triangle::intersectRay rejects such determinants,
// Primitives.h:41-42): every operand whose result is used already lies above the fast path's lower bound 2^-125, so
// one compare (the upper bound; it also sends NaN to the IEEE sequence) replaces two.
PTM_HD float rcp_if_above_1em7(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    float out = __builtin_fmaf(e, r0, r0);
    const bool inRange = __builtin_fabsf(x) < 8.5070592e37f /* 2^126 */;
    if (__builtin_expect(!allLanes(lanes(inRange)), 0)) out = inRange ? out : 1.0f / x;
    return out;
#else
    return 1.0f / x;
#endif
}
// The fast path alone, for a caller that has ALREADY established 2^-125 <= |x| < 2^126 for every operand whose result
// it uses (the closest-hit triangle loop proves the upper bound once per query from |d| and the scene's largest
// |e1| |e2|, and discards results with |x| <= 1e-7 like rcp_if_above_1em7's callers). Same three instructions, same bits.
PTM_HD float rcp_in_range(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
#else
    return 1.0f / x;
#endif
}
PTM_HD uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
PTM_HD float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
PTM_HD float inf() { return u2f(0x7f800000u); }
PTM_HD float qnan() { return u2f(0x7fc00000u); }

// div(a, b) = a / b, correctly rounded (IEEE) on both sides. Device fast path (Markstein): with r = RN(1/b) from the
// rcp fast path, q0 = a*r, q = fma(fma(-b, q0, a), r, q0). Bit-identical to the IEEE quotient for ALL 2^46 pairs of
// float32 mantissas (tests/csrc/math_exhaustive.hip; power-of-two scaling is exact, so mantissas decide), as long as
// nothing under- or overflows: guarded to |a|, |b| in [2^-60, 2^60); everything else (zeros, infinities, NaNs, extreme
// exponents) takes the IEEE sequence through a wave-uniform escape.
#if defined(__HIP_DEVICE_COMPILE__)
PTM_HD bool div_in_range(float x) {
    const float ax = __builtin_fabsf(x);
    return ax >= 8.6736174e-19f /* 2^-60 */ && ax < 1.1529215e18f /* 2^60 */;
}
PTM_HD unsigned long long div_range_lanes(float x) {  // the same test as a wave mask
    const float ax = __builtin_fabsf(x);
    return lanes(ax >= 8.6736174e-19f) & lanes(ax < 1.1529215e18f);
}
PTM_HD float rcp_core(float b) {  // only for operands already known to be in range
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
PTM_HD float div_core(float a, float b, float r) {
    const float q0 = a * r;
    const float rem = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(rem, r, q0);
}
#endif
PTM_HD float div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = div_core(a, b, rcp_core(b));
    if (__builtin_expect(!allLanes(div_range_lanes(a) & div_range_lanes(b)), 0)) q = (div_in_range(a) && div_in_range(b)) ? q : a / b;
    return q;
#else
    return a / b;
#endif
}
// three numerators over one denominator (glm's vec3 / float is three divisions): one reciprocal serves all
PTM_HD void div3(float ax, float ay, float az, float b, float& qx, float& qy, float& qz) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = rcp_core(b);
    qx = div_core(ax, b, r);
    qy = div_core(ay, b, r);
    qz = div_core(az, b, r);
    if (__builtin_expect(!allLanes(div_range_lanes(b) & div_range_lanes(ax) & div_range_lanes(ay) & div_range_lanes(az)), 0)) {
        const bool inRange = div_in_range(b) && div_in_range(ax) && div_in_range(ay) && div_in_range(az);
        qx = inRange ? qx : ax / b;
        qy = inRange ? qy : ay / b;
        qz = inRange ? qz : az / b;
    }
#else
    qx = ax / b;
    qy = ay / b;
    qz = az / b;
#endif
}
// glm::max(a,b) = (a < b) ? b : a ; glm::min(a,b) = (b < a) ? b : a  (ordered compares, NaN-stable)
PTM_HD float max(float a, float b) { return (a < b) ? b : a; }
PTM_HD float min(float a, float b) { return (b < a) ? b : a; }
PTM_HD float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }

// ---------------------------------------------------------------------------------------------
// sin & cos together. Cody–Waite reduction by pi/2 in three fma steps, Cephes sinf/cosf kernels.
// Valid (<= 2 ulp) for |x| <= 1e4; NaN for NaN/inf/huge.
// ---------------------------------------------------------------------------------------------
PTM_HD void sincos(float x, float& s, float& c) {
    if (!(abs(x) <= 1.0e4f)) {
        s = qnan();
        c = qnan();
        return;
    }
    float fn = __builtin_rintf(x * 0.6366197466850281f);
    int n = (int)fn;
    float r = fma(fn, -1.5707963705062866f, x);
    r = fma(fn, 4.371138828673793e-08f, r);
    r = fma(fn, 1.7151245100058819e-15f, r);
    float z = r * r;
    float sp = fma(fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    sp = fma(sp * z, r, r);
    float cp = fma(fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    cp = fma(cp * z, z, fma(-0.5f, z, 1.0f));
    float ss = (n & 1) ? cp : sp;
    float cc = (n & 1) ? sp : cp;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
PTM_HD float tan(float x) {
    float s, c;
    sincos(x, s, c);
    return div(s, c);
}

// atan, Cephes atanf reduction (tan 3pi/8, tan pi/8) + degree-4 odd polynomial.
PTM_HD float atan(float x) {
    if (x != x) return x;
    float ax = abs(x);
    float y, t;
    if (ax > 2.414213562373095f) {
        y = 1.5707963267948966f;
        t = -rcp(ax);
    } else if (ax > 0.4142135623730950f) {
        y = 0.7853981633974483f;
        t = div(ax - 1.0f, ax + 1.0f);
    } else {
        y = 0.0f;
        t = ax;
    }
    float z = t * t;
    float p = fma(fma(fma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    y = y + fma(p * z, t, t);
    return (x < 0.0f) ? -y : y;
}

// natural log, Cephes logf. log(0) = -inf, log(<0) = NaN, denormals handled.
PTM_HD float log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return -inf();
    if (x == inf()) return x;
    int e = 0;
    if (x < 1.17549435e-38f) {
        x = x * 8388608.0f;
        e = -23;
    }
    uint32_t b = f2u(x);
    e += (int)((b >> 23) & 0xffu) - 126;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);  // [0.5, 1)
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fma(p, m, -1.1514610310e-1f);
    p = fma(p, m, 1.1676998740e-1f);
    p = fma(p, m, -1.2420140846e-1f);
    p = fma(p, m, 1.4249322787e-1f);
    p = fma(p, m, -1.6668057665e-1f);
    p = fma(p, m, 2.0000714765e-1f);
    p = fma(p, m, -2.4999993993e-1f);
    p = fma(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fma(-2.12194440e-4f, fe, y);
    y = fma(-0.5f, z, y);
    float r = m + y;
    return fma(0.693359375f, fe, r);
}

// exp, Cephes expf. Results below FLT_MIN flush to +0 (deterministic on both sides).
PTM_HD float exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return inf();
    if (x < -87.33654475055310f) return 0.0f;
    float fn = __builtin_rintf(x * 1.4426950216293335f);
    int n = (int)fn;
    float r = fma(fn, -0.693359375f, x);
    r = fma(fn, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fma(p, r, 1.3981999507e-3f);
    p = fma(p, r, 8.3334519073e-3f);
    p = fma(p, r, 4.1665795894e-2f);
    p = fma(p, r, 1.6666665459e-1f);
    p = fma(p, r, 5.0000001201e-1f);
    float y = fma(p, z, r) + 1.0f;
    // scale by 2^n in two exact steps so n = 128 / n = -126 stay in range
    int h = n / 2;
    float s1 = u2f((uint32_t)(h + 127) << 23);
    float s2 = u2f((uint32_t)((n - h) + 127) << 23);
    return (y * s1) * s2;
}

// pow for x >= 0 (the only use: gamma 1/2.2 and the Phong lobe exponent). x<0 -> NaN.
PTM_HD float pow(float x, float y) {
    if (x != x || y != y) return qnan();
    if (x < 0.0f) return qnan();
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : inf();
    if (x == 1.0f) return 1.0f;
    return exp(y * log(x));
}

}  // namespace ptm

// ---------------------------------------------------------------------------------------------
// Vector layer. vec3/quat ARE the boundary PODs (ptss_types.h), so scene records can be used
// in place. Import with `using namespace ptv;` (scalar ptm:: functions stay qualified so they can
// never be confused with libm / device-libs overloads).
// ---------------------------------------------------------------------------------------------
namespace ptv {
using vec3 = ::ptss_vec3;  // x,y,z
using quat = ::ptss_quat;  // glm memory order x,y,z,w; glm ctor order (w,x,y,z)


PTM_HD vec3 v3(float x, float y, float z) { return vec3{x, y, z}; }
PTM_HD vec3 v3(float s) { return vec3{s, s, s}; }
PTM_HD vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTM_HD vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTM_HD vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
PTM_HD vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PTM_HD vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
PTM_HD vec3 operator*(float s, vec3 a) { return vec3{a.x * s, a.y * s, a.z * s}; }
PTM_HD vec3 operator/(vec3 a, float s) {
    vec3 q;
    ptm::div3(a.x, a.y, a.z, s, q.x, q.y, q.z);
    return q;
}
// o + v*s with one rounding per component (the contraction nvcc applies to `o + d * t`)
PTM_HD vec3 madd(vec3 v, float s, vec3 o) { return vec3{ptm::fma(v.x, s, o.x), ptm::fma(v.y, s, o.y), ptm::fma(v.z, s, o.z)}; }
// o + a*b component-wise, one rounding
PTM_HD vec3 madd(vec3 a, vec3 b, vec3 o) { return vec3{ptm::fma(a.x, b.x, o.x), ptm::fma(a.y, b.y, o.y), ptm::fma(a.z, b.z, o.z)}; }

PTM_HD float dot(vec3 a, vec3 b) { return ptm::fma(a.z, b.z, ptm::fma(a.y, b.y, a.x * b.x)); }
PTM_HD vec3 cross(vec3 a, vec3 b) {
    return vec3{ptm::fma(a.y, b.z, -(a.z * b.y)), ptm::fma(a.z, b.x, -(a.x * b.z)), ptm::fma(a.x, b.y, -(a.y * b.x))};
}
// glm::normalize(v) = v * inversesqrt(dot(v,v))
PTM_HD vec3 normalize(vec3 v) { return v * ptm::rcp(ptm::sqrt(dot(v, v))); }
PTM_HD float length(vec3 v) { return ptm::sqrt(dot(v, v)); }

PTM_HD quat q4(float w, float x, float y, float z) { return quat{x, y, z, w}; }
// glm::normalize(quat): identity when the length is not positive
PTM_HD quat normalize(quat q) {
    float len = ptm::sqrt(ptm::fma(q.w, q.w, ptm::fma(q.z, q.z, ptm::fma(q.y, q.y, q.x * q.x))));
    if (len <= 0.0f) return q4(1, 0, 0, 0);
    float inv = ptm::rcp(len);
    return q4(q.w * inv, q.x * inv, q.y * inv, q.z * inv);
}
// glm quat * vec3:  v + 2w (u x v) + 2 (u x (u x v))
PTM_HD vec3 rotate(quat q, vec3 v) {
    vec3 u = v3(q.x, q.y, q.z);
    vec3 uv = cross(u, v);
    vec3 uuv = cross(u, uv);
    uv = uv * (2.0f * q.w);
    uuv = uuv * 2.0f;
    return (v + uv) + uuv;
}
// Hamilton product p*q (glm operator*)
PTM_HD quat mul(quat p, quat q) {
    return q4(p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z,
              p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y,
              p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z,
              p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x);
}

}  // namespace ptv
