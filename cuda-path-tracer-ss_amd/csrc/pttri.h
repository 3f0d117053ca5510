// pttri.h — the arithmetic of Triangle::intersectRay (Primitives.h:25-83) as straight-line algebra, written once for
// the gfx950 kernels (ptss_kernels.hip) and for a host probe (host_capi.cpp, tests/test_triangle_forms.py).
//
// The reference evaluates, per triangle (e1 = v1 - v0, e2 = v2 - v0 hoisted to the host, same subtraction):
//     q = d x e2;  det = e1 . q;  inv = 1 / det;  s = o - v0;  r = s x e1;  dist = (e2 . r) inv
//     b1 = (s . q) inv;  b2 = (d . r) inv;  b0 = 1 - (b1 + b2)
// with dot(a, b) = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)) and cross(a, b).x = fma(a.y, b.z, -(a.z * b.y)) ... (ptmath.h).
//
// EDGE CLASSES. Every preset's walls and light panels are rectangles whose edges run along one coordinate axis
// (Scene.cpp:323-370; of configs[2]'s 16 triangles 12 have two such edges and the other 4 one). The host records per edge
// which components are EXACT zeros (class 0: none known; 1 / 2 / 3: only x / y / z may be non-zero — csrc/ptss_api.hip
// packScene, compared with == 0.0f), and the class forms below leave out the products with those components. Why that is
// the same arithmetic: for finite x, x * (+-0) = +-0 exactly, fma(a, b, +-0) = RN(a b) and y + (+-0) = y for y != 0, so
//   (1) every NON-ZERO intermediate value is bit-identical in both forms;
//   (2) an intermediate that is zero in one form is zero in the other — possibly of the other sign.
// A zero's sign never reaches a decision: det = +-0 is rejected by |det| <= 1e-7 (Primitives.h:41), dist = +-0 by
// dist <= 0 (:52), a weight of +-0 is "not < 0" either way (:59-62), and b1 + b2 with one or two zero addends gives the
// same b0 (1 - (+-0) = 1). So both forms accept the same triangles at the same (non-zero) distance with the same non-zero
// weights; only a kept weight that is EXACTLY zero (a hit exactly on an edge of the triangle) may carry the other sign,
// and the caller re-evaluates such a hit with the general form (closestHit: `zeroWeight`), so that even the sign of a zero
// normal component downstream is the reference's. Preconditions, established by the callers: the flagged components are
// exact zeros (host), and everything they are multiplied with is FINITE — d, s = o - v0, q = d x e2 and r = s x e1 —, or the
// general form's inf * 0 = NaN would be lost: the host records classes only for bounded geometry (every |coordinate| <= 1e15,
// SceneLayout::triClassed) and the kernel tests |d|^2 < 2^30 and |o|^2 < 2^100 once per query, which keeps |q| and |r| below
// 1e32. tests/test_triangle_forms.py enumerates the sign-of-zero, large and tiny cases on the host build of this file and
// holds counter-examples from outside the domain; the GPU parity suite runs the kernels.
#pragma once
#include "ptmath.h"

namespace pttri {
using namespace ptv;

// ---- what a class says -------------------------------------------------------------------------------------------------
constexpr bool edgeZero(int cls, int comp) { return cls != 0 && comp != cls - 1; }   // component comp of the edge is an exact zero
constexpr bool crossZero(int cls, int comp) { return cls != 0 && comp == cls - 1; }  // component comp of (a x edge) is a zero
// class of an edge vector as the host sees it (== 0.0f matches -0.0f too)
PTM_HD int edgeClass(vec3 e) {
    const bool zx = e.x == 0.0f, zy = e.y == 0.0f, zz = e.z == 0.0f;
    if (zy && zz && !zx) return 1;
    if (zx && zz && !zy) return 2;
    if (zx && zy && !zz) return 3;
    return 0;
}
// Packed as 4 bits per triangle: class(e1) * 4 + class(e2); both edges along the SAME axis (a degenerate triangle, det = 0)
// keeps only e1's class, so that 13 of the 16 codes occur.
PTM_HD int triangleClass(vec3 e1, vec3 e2) {
    const int c1 = edgeClass(e1);
    int c2 = edgeClass(e2);
    if (c1 != 0 && c1 == c2) c2 = 0;
    return c1 * 4 + c2;
}

// ---- zero-aware pieces: exactly ptv::cross / ptv::dot with the flagged products left out ---------------------------------
template <bool kZ1, bool kZ2>   // fma(a1, b1, -(a2 * b2)) where b1 (kZ1) / b2 (kZ2) is a known zero
PTM_HD float crossTerm(float a1, float b1, float a2, float b2) {
    if constexpr (kZ1 && kZ2) return 0.0f;
    else if constexpr (kZ2) return a1 * b1;
    else if constexpr (kZ1) return -(a2 * b2);
    else return ptm::fma(a1, b1, -(a2 * b2));
}
template <int kCls>   // a x e for an edge e of class kCls
PTM_HD vec3 crossEdge(vec3 a, vec3 e) {
    return vec3{crossTerm<edgeZero(kCls, 2), edgeZero(kCls, 1)>(a.y, e.z, a.z, e.y),
                crossTerm<edgeZero(kCls, 0), edgeZero(kCls, 2)>(a.z, e.x, a.x, e.z),
                crossTerm<edgeZero(kCls, 1), edgeZero(kCls, 0)>(a.x, e.y, a.y, e.x)};
}
template <bool kZ0, bool kZ1, bool kZ2>   // fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)) without the terms flagged zero
PTM_HD float dotSkip(vec3 a, vec3 b) {
    if constexpr (kZ0 && kZ1 && kZ2) return 0.0f;
    else if constexpr (kZ0 && kZ1) return a.z * b.z;
    else if constexpr (kZ0 && kZ2) return a.y * b.y;
    else if constexpr (kZ1 && kZ2) return a.x * b.x;
    else if constexpr (kZ0) return ptm::fma(a.z, b.z, a.y * b.y);
    else if constexpr (kZ1) return ptm::fma(a.z, b.z, a.x * b.x);
    else if constexpr (kZ2) return ptm::fma(a.y, b.y, a.x * b.x);
    else return dot(a, b);
}

// ---- the test in two stages (the kernels exit between them when no lane of the wave can still be hit) -------------------
struct Head {   // up to the distance test, Primitives.h:34-52
    vec3 q, s, r;
    float det, inv, dist;
};
// kC1 / kC2: classes of e1 / e2. kPrimary: s, r and e2 . r come precomputed (every eye ray starts at the camera:
// primaryPrepKernel evaluates them with the general form, so r's flagged components are exact zeros there as well).
// the part that depends on the ray's ORIGIN only (two shadow segments leaving one surface point share it: pairAnyHit)
struct OriginPart {
    vec3 s, r;
    float e2r;
};
template <int kC1, int kC2>
PTM_HD OriginPart originPart(vec3 v0, vec3 e1, vec3 e2, vec3 o) {
    OriginPart p;
    p.s = o - v0;
    p.r = crossEdge<kC1>(p.s, e1);
    p.e2r = dotSkip<edgeZero(kC2, 0) || crossZero(kC1, 0), edgeZero(kC2, 1) || crossZero(kC1, 1), edgeZero(kC2, 2) || crossZero(kC1, 2)>(e2, p.r);
    return p;
}
// ... and the part that depends on the direction, given the origin part
template <int kC1, int kC2>
PTM_HD Head headFrom(const OriginPart& p, vec3 e1, vec3 e2, vec3 d) {
    Head h;
    h.q = crossEdge<kC2>(d, e2);
    h.det = dotSkip<edgeZero(kC1, 0) || crossZero(kC2, 0), edgeZero(kC1, 1) || crossZero(kC2, 1), edgeZero(kC1, 2) || crossZero(kC2, 2)>(e1, h.q);
    h.inv = ptm::rcp_in_range(h.det);   // 1 / det, :44 — the caller has bounded |det| < 2^126 and discards results with |det| <= 1e-7
    h.s = p.s;
    h.r = p.r;
    h.dist = p.e2r * h.inv;
    return h;
}
template <int kC1, int kC2, bool kPrimary>
PTM_HD Head head(vec3 v0, vec3 e1, vec3 e2, vec3 ps, vec3 pr, float pe2r, vec3 o, vec3 d) {
    if constexpr (kPrimary) return headFrom<kC1, kC2>(OriginPart{ps, pr, pe2r}, e1, e2, d);
    else return headFrom<kC1, kC2>(originPart<kC1, kC2>(v0, e1, e2, o), e1, e2, d);
}
template <int kC1, int kC2>
PTM_HD void weights(const Head& h, vec3 d, float& b0, float& b1, float& b2) {   // :55-64
    b1 = dotSkip<crossZero(kC2, 0), crossZero(kC2, 1), crossZero(kC2, 2)>(h.s, h.q) * h.inv;
    b2 = dotSkip<crossZero(kC1, 0), crossZero(kC1, 1), crossZero(kC1, 2)>(d, h.r) * h.inv;
    b0 = 1.0f - (b1 + b2);
}

// the reference's verdict from the pieces (host probe and documentation; the kernels form the same predicates as wave masks)
PTM_HD bool passesHead(const Head& h, float limit) { return !(ptm::abs(h.det) <= 1e-7f) && !(h.dist <= 0.0f) && !(h.dist > limit); }
PTM_HD bool passesWeights(float b0, float b1, float b2) { return !(b0 < 0) && !(b1 < 0) && !(b2 < 0); }

}  // namespace pttri
