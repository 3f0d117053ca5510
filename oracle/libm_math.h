// libm_math.h — the SAME interface as cuda-path-tracer-ss_amd/csrc/ptmath.h (namespaces ptm / ptv), written with the host's
// libm float functions and the plainest vector arithmetic. TEST INFRASTRUCTURE.
//
// Why: oracle/oracle.cpp normally includes the product's ptmath.h, so a mistake in ptm::sin / atan / log / exp / pow or in a
// pinned fma chain would be wrong on BOTH sides of every array_equal parity test. Building the oracle a second time
// against this header (oracle/build.py -> _build/liboracle_libm.so, -DORACLE_LIBM_MATH) gives a renderer that shares no
// arithmetic with the product; tests/test_oracle_libm.py renders the same scenes with both and requires the images to
// agree statistically (they differ in the last ulp, hence in individual branch decisions, hence per sample — but not in
// the mean). Follows the glm / CUDA-libm call sites listed at the top of ptmath.h.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "ptss_types.h"

#define PTM_HD inline

namespace ptm {
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618f;
constexpr float kRayBump = 1e-4f;
constexpr float kGamma = (1 / 2.2f);
inline float fma(float a, float b, float c) { return a * b + c; }
inline float abs(float x) { return fabsf(x); }
inline float sqrt(float x) { return sqrtf(x); }
inline float rcp(float x) { return 1.0f / x; }
inline float rcp_if_above_1em7(float x) { return 1.0f / x; }
inline float rcp_in_range(float x) { return 1.0f / x; }
inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
inline float inf() { return INFINITY; }
inline float qnan() { return NAN; }
inline float div(float a, float b) { return a / b; }
inline void div3(float ax, float ay, float az, float b, float& qx, float& qy, float& qz) { qx = ax / b; qy = ay / b; qz = az / b; }
inline float max(float a, float b) { return (a < b) ? b : a; }
inline float min(float a, float b) { return (b < a) ? b : a; }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline void sincos(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
inline float tan(float x) { return tanf(x); }
inline float atan(float x) { return atanf(x); }
inline float log(float x) { return logf(x); }
inline float exp(float x) { return expf(x); }
inline float pow(float x, float y) { return powf(x, y); }
}  // namespace ptm

namespace ptv {
using vec3 = ::ptss_vec3;
using quat = ::ptss_quat;
inline vec3 v3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 v3(float s) { return vec3{s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return vec3{a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return vec3{a.x / s, a.y / s, a.z / s}; }
inline vec3 madd(vec3 v, float s, vec3 o) { return vec3{v.x * s + o.x, v.y * s + o.y, v.z * s + o.z}; }
inline vec3 madd(vec3 a, vec3 b, vec3 o) { return vec3{a.x * b.x + o.x, a.y * b.y + o.y, a.z * b.z + o.z}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
inline quat q4(float w, float x, float y, float z) { return quat{x, y, z, w}; }
inline quat normalize(quat q) {
    const float len = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    if (len <= 0.0f) return q4(1, 0, 0, 0);
    const float inv = 1.0f / len;
    return q4(q.w * inv, q.x * inv, q.y * inv, q.z * inv);
}
inline vec3 rotate(quat q, vec3 v) {  // glm quat * vec3
    const vec3 u = v3(q.x, q.y, q.z);
    vec3 uv = cross(u, v);
    vec3 uuv = cross(u, uv);
    uv = uv * (2.0f * q.w);
    uuv = uuv * 2.0f;
    return (v + uv) + uuv;
}
inline quat mul(quat p, quat q) {
    return q4(p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z, p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y,
              p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z, p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x);
}
}  // namespace ptv
