// ptss_kernels.hip — the hot path as hand-written HIP for gfx950 (CDNA4, 64-lane waves).
//
// Kernels (reference kernels they replace, paths relative to /root/reference/CudaTracer/):
//   rngInitKernel   <- curandSetupKernel            CudaTracer.cu:22-29
//   clearKernel     <- clearPixels                  CudaTracer.cu:31-49
//   bounceKernel<first> <- computeEyeRaysKernel     CudaTracer.cu:51-61, 321-343 (fused into bounce 0)
//   bounceKernel    <- pathTraceKernel + thrust::partition + the per-ray part of writeToPixelsKernel
//                                                   CudaTracer.cu:106-206, :629, :63-104
//   flushKernel     <- writeToPixelsKernel for rays still alive when the loop guard stops the frame
//                                                   CudaTracer.cu:622, :63-104
//
// Design: one ray per lane; ray state in SoA planes (coalesced 256-B wave accesses), pools cut
// into kShards regions with one live-ray counter each; the whole scene staged once per workgroup into LDS
// and read by broadcast; divergence is attacked at WAVE level, without barriers: sphere hits are resolved
// from per-lane candidate bit masks, triangle tests exit wave-uniformly, and the shadow rays of two lights
// at a time are regrouped densely through a wave-private LDS queue; live rays are compacted by the wave
// that traced them (64-bit ballot + popcount lane rank + one returning atomic per wave on the shard's
// device-resident counter, issued before the tone-mapping of the finished lanes so its round trip hides),
// so the host never reads a ray count inside a frame; a path that ends tone-maps into the integer
// accumulator right there and parks its XORWOW state in the per-pixel home record. Bounce 0 makes its own
// eye rays. Reciprocals, square roots and divisions use hardware approximation + one fma correction where
// that is proven bit-identical to IEEE (ptmath.h). No MFMA: there is no dense contraction in this path.
//
// Arithmetic mirrors oracle/oracle.cpp operation for operation (ptmath.h; -ffp-contract=off);
// every restructuring below is argued exact where it is made.
#include "ptss_device.h"
#include "ptquant.h"
#include "pttri.h"

using namespace ptv;

namespace ptss {
namespace {

__device__ __forceinline__ float asF(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t asU(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ vec3 xyz(float4 v) { return vec3{v.x, v.y, v.z}; }
// orders this wave's LDS traffic for the compiler; within one wave the LDS executes in order
// "does any lane of the wave say yes": a ballot compared with zero stays in scalar registers (s_and / s_cmp / s_cbranch);
// hipcc's __any() round-trips the mask through a VGPR (v_cndmask + v_cmp) — two VALU instructions per triangle test
__device__ __forceinline__ bool waveAny(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// "every active lane says yes" (p must be one direct compare, see maskOf)
__device__ __forceinline__ bool waveAll(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
__device__ __forceinline__ void waveLdsFence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct PixelCoord {
    int x, gy;
    uint32_t globalIndex;
};

__device__ __forceinline__ PixelCoord locate(const TileMap& t, uint32_t local) {
    const int lx = (int)(local % (uint32_t)t.width);
    const int ly = (int)(local / (uint32_t)t.width);
    const int band = ly / t.bandRows, within = ly % t.bandRows;
    PixelCoord p;
    p.x = lx;
    p.gy = (band * t.world + t.rank) * t.bandRows + within;
    p.globalIndex = (uint32_t)p.gy * (uint32_t)t.width + (uint32_t)lx;
    return p;
}

// Ray::pixelOffset as carried by a ray: local pixel in the low 26 bits, sample lane (0..S-1, S <= 64) above.
// S = cfg.samplesPerPass independent random streams per pixel are traced per pass (1 = the reference).
constexpr uint32_t kLaneShift = 26;
constexpr uint32_t kPixMask = (1u << kLaneShift) - 1u;
__device__ __forceinline__ uint32_t pixOf(uint32_t packed) { return packed & kPixMask; }
__device__ __forceinline__ uint32_t laneOf(uint32_t packed) { return packed >> kLaneShift; }

struct RayRegs {
    vec3 o, d, L0, T;
    uint32_t pix;
    ptrng::State rng;
    bool active;
};

// ---- Ray pool addressing (ptss_device.h "ray pools"): a shard's region is a row of TILE BLOCKS, one per kBlock rays, each
// holding the kRayPlanes planes of its rays back to back: word (tile t, plane p, lane w) sits at (t * kRayPlanes + p) *
// kBlock + w. A tile of a workgroup is one block: its base is wave-uniform (scalar registers), the lane offset is
// threadIdx.x and the plane offset a compile-time constant, so a plane access needs no vector address arithmetic at all
// (the plane-major layout of round 1 spent a v_add_u32 + v_lshl_add_u64 per plane — 38 per tile, and both are half-rate
// instructions on gfx950: tools/microbench/vgpr_banks.hip). Survivors are stored at region slot `slot`: block
// slot / kBlock, lane slot % kBlock — one multiply-add per ray. Every access is still a 256-B contiguous wave transaction.
__device__ __forceinline__ const float* tileBlock(const float* __restrict__ region, uint32_t firstSlot /* multiple of kBlock */) {
    return region + (size_t)(firstSlot / kBlock) * (kRayPlanes * kBlock);
}
__device__ __forceinline__ uint32_t slotWord(uint32_t slot) {  // word offset of (slot, plane 0) inside the region
    return (slot / kBlock) * (uint32_t)(kRayPlanes * kBlock) + (slot % kBlock);
}

// One word of a block: scalar base + (32-bit lane byte offset, zero-extended) + compile-time plane offset — the form
// global_load/store take as `saddr + voffset + imm` (no 64-bit vector address pair per group of planes).
// kCoherent (the one-launch-per-frame kernel, frameKernel): the word was written, or will be read, by ANOTHER workgroup of the
// same launch — relaxed agent-scope accesses (global_load / global_store ... sc1: past the CU's L1, written through), the
// payload half of the sc1 hand-off of MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility".
template <bool kCoherent = false>
__device__ __forceinline__ float ldPlane(const float* __restrict__ block, uint32_t laneBytes, int plane) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(block) + (size_t)laneBytes + (size_t)plane * (kBlock * sizeof(float)));
    if constexpr (kCoherent) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool kCoherent = false>
__device__ __forceinline__ void stPlane(float* __restrict__ region, uint32_t wordBytes, int plane, float v) {
    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(region) + (size_t)wordBytes + (size_t)plane * (kBlock * sizeof(float)));
    if constexpr (kCoherent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// A tile fetches a ray's planes in the order it needs them, so that no plane occupies registers before
// its consumer runs: origin + direction for the closest-hit loops, the XORWOW state for the light samples, radiance /
// throughput / pixel for the update at the end. `block` = the tile's block (wave-uniform), w = the ray's lane in it.
template <bool kCoherent = false>
__device__ __forceinline__ void loadRayGeometry(const float* __restrict__ block, uint32_t w, RayRegs& r) {
    const uint32_t b = w * 4u;
    r.o = vec3{ldPlane<kCoherent>(block, b, kOx), ldPlane<kCoherent>(block, b, kOy), ldPlane<kCoherent>(block, b, kOz)};
    r.d = vec3{ldPlane<kCoherent>(block, b, kDx), ldPlane<kCoherent>(block, b, kDy), ldPlane<kCoherent>(block, b, kDz)};
    r.active = true;
}
template <bool kCoherent = false>
__device__ __forceinline__ void loadRayRng(const float* __restrict__ block, uint32_t w, RayRegs& r) {
    const uint32_t b = w * 4u;
    r.rng.v[0] = asU(ldPlane<kCoherent>(block, b, kR0));
    r.rng.v[1] = asU(ldPlane<kCoherent>(block, b, kR1));
    r.rng.v[2] = asU(ldPlane<kCoherent>(block, b, kR2));
    r.rng.v[3] = asU(ldPlane<kCoherent>(block, b, kR3));
    r.rng.v[4] = asU(ldPlane<kCoherent>(block, b, kR4));
    r.rng.d = asU(ldPlane<kCoherent>(block, b, kRd));
}
template <bool kCoherent = false>
__device__ __forceinline__ void loadRayRadiance(const float* __restrict__ block, uint32_t w, RayRegs& r) {
    const uint32_t b = w * 4u;
    r.L0 = vec3{ldPlane<kCoherent>(block, b, kL0x), ldPlane<kCoherent>(block, b, kL0y), ldPlane<kCoherent>(block, b, kL0z)};
    r.T = vec3{ldPlane<kCoherent>(block, b, kTx), ldPlane<kCoherent>(block, b, kTy), ldPlane<kCoherent>(block, b, kTz)};
    r.pix = asU(ldPlane<kCoherent>(block, b, kPix));
}
template <bool kCoherent = false>
__device__ __forceinline__ void loadRay(const float* __restrict__ block, uint32_t w, RayRegs& r) {
    loadRayGeometry<kCoherent>(block, w, r);
    loadRayRng<kCoherent>(block, w, r);
    loadRayRadiance<kCoherent>(block, w, r);
}

// the ray goes to region slot `slot` (its word in plane 0 of its block: slotWord)
template <bool kCoherent = false>
__device__ __forceinline__ void storeRay(float* __restrict__ region, uint32_t slot, const RayRegs& r) {
    const uint32_t b = slotWord(slot) * 4u;   // < 2^32: ptss_create bounds a region's bytes
    stPlane<kCoherent>(region, b, kOx, r.o.x);   stPlane<kCoherent>(region, b, kOy, r.o.y);   stPlane<kCoherent>(region, b, kOz, r.o.z);
    stPlane<kCoherent>(region, b, kDx, r.d.x);   stPlane<kCoherent>(region, b, kDy, r.d.y);   stPlane<kCoherent>(region, b, kDz, r.d.z);
    stPlane<kCoherent>(region, b, kL0x, r.L0.x); stPlane<kCoherent>(region, b, kL0y, r.L0.y); stPlane<kCoherent>(region, b, kL0z, r.L0.z);
    stPlane<kCoherent>(region, b, kTx, r.T.x);   stPlane<kCoherent>(region, b, kTy, r.T.y);   stPlane<kCoherent>(region, b, kTz, r.T.z);
    stPlane<kCoherent>(region, b, kPix, asF(r.pix));
    stPlane<kCoherent>(region, b, kR0, asF(r.rng.v[0]));
    stPlane<kCoherent>(region, b, kR1, asF(r.rng.v[1]));
    stPlane<kCoherent>(region, b, kR2, asF(r.rng.v[2]));
    stPlane<kCoherent>(region, b, kR3, asF(r.rng.v[3]));
    stPlane<kCoherent>(region, b, kR4, asF(r.rng.v[4]));
    stPlane<kCoherent>(region, b, kRd, asF(r.rng.d));
}

// ---- Sphere::intersectRay, Primitives.h:107-175. sp = {centre, radius^2}. ---------------------
// The reference's first exit, `discriminent < 0` (Primitives.h:117-118), is what the candidate masks below decide for up to
// 32 spheres at a time (shiftInSphere); sphereTest is the whole test, for the candidates. Both evaluate b, c and the
// discriminant with the same operations.
// Returns the accepted distance in t; `limit` is the running `distance`.
__device__ __forceinline__ bool sphereTest(float4 sp, vec3 o, vec3 d, float limit, float& t) {
    const vec3 v = o - xyz(sp);
    const float b = dot(d, v) * 2;
    const float c = dot(v, v) - sp.w;
    float disc = (b * b) - 4 * c;
    if (disc < 0) return false;
    disc = ptm::sqrt(disc);
    float t0 = (-b + disc) * 0.5f;
    float t1 = (-b - disc) * 0.5f;
    if (t0 < 0 && t1 < 0) return false;
    if (t0 > t1) {
        const float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    const float cand = (t0 < 0) ? t1 : t0;
    if (cand > limit) return false;
    t = cand;
    return true;
}

// ---- Triangle::intersectRay, Primitives.h:25-83, with the per-lane exits replaced by ONE
// wave-uniform exit: every lane computes det, 1/det and dist (selects instead of divergent
// branches: no exec-mask bookkeeping, and the straight-line code lets the scheduler overlap the
// long division chain with the cross products); the barycentric part runs only if some lane of
// the wave passed both early tests. `live` marks lanes whose result matters. Same operations on
// the same values as the reference for every lane that the reference would carry that far;
// lanes it would have dropped compute values that are discarded. -----------------------------------
struct TriHit {
    bool hit;
    unsigned long long hitMask;  // the same verdicts as a wave mask
    float dist, w0, w1, w2;
};

struct TriRows {  // one staged triangle: {v0, bits(materialIdx)}, {e1, 0}, {e2, 0}
    float4 a, b, c;
};
// The tests use three of a row's four words, and hipcc narrows each fetch to ds_read_b96 — for these broadcast reads the
// faster form (8.4 against 14 SIMD-cycles per wave-read for ds_read_b128, tools/microbench/loops.hip: the LDS-to-VGPR
// return path moves bytes, and 768 are fewer than 1,024).
__device__ __forceinline__ float4 loadRow16(const float4* p) { return *p; }
__device__ __forceinline__ TriRows loadTri(const float4* tr) { return TriRows{loadRow16(tr), loadRow16(tr + 1), loadRow16(tr + 2)}; }
// the camera-origin test (triangleTestPrimary) never looks at v0
__device__ __forceinline__ TriRows loadTriEdges(const float4* tr) { return TriRows{float4{0, 0, 0, 0}, loadRow16(tr + 1), loadRow16(tr + 2)}; }

__device__ __forceinline__ float triRcp(float det) { return ptm::rcp_if_above_1em7(det); }

// Lane predicates travel as 64-bit wave masks (one v_cmp each, combined with scalar ANDs, carried over the wave-uniform
// branch in SGPRs, turned back into a lane predicate for free by inverse_ballot). As bools they made hipcc round-trip
// through a VGPR — v_cndmask + v_cmp — every time a compound condition met a ballot: twice per triangle.
__device__ __forceinline__ unsigned long long maskOf(bool directCompare) { return __builtin_amdgcn_ballot_w64(directCompare); }

__device__ __forceinline__ TriHit triangleTest(const TriRows& tr, vec3 o, vec3 d, float limit, unsigned long long liveMask) {
    const vec3 v0 = xyz(tr.a), e1 = xyz(tr.b), e2 = xyz(tr.c);
    const vec3 q = cross(d, e2);
    const float det = dot(e1, q);
    const float inverseDet = triRcp(det);  // 1 / det, Primitives.h:44; unused when |det| <= 1e-7
    const vec3 s = o - v0;
    const vec3 r = cross(s, e1);
    const float dist = dot(e2, r) * inverseDet;
    // pass = live && !(|det| <= 1e-7) && !(dist <= 0 || dist > limit), Primitives.h:41-42, :51-52
    const unsigned long long passMask = liveMask & maskOf(!(ptm::abs(det) <= 1e-7f)) & maskOf(!(dist <= 0.0f)) & maskOf(!(dist > limit));
    TriHit h;
    h.hit = false;
    h.hitMask = 0ull;
    h.dist = dist;
    h.w0 = h.w1 = h.w2 = 0;
    if (passMask != 0ull) {
        const float b1 = dot(s, q) * inverseDet;
        const float b2 = dot(d, r) * inverseDet;
        const float b0 = 1.0f - (b1 + b2);
        h.hitMask = passMask & maskOf(!(b0 < 0)) & maskOf(!(b1 < 0)) & maskOf(!(b2 < 0));
        h.hit = __builtin_amdgcn_inverse_ballot_w64(h.hitMask);
        h.w0 = b0;
        h.w1 = b1;
        h.w2 = b2;
    }
    return h;
}

// ---- The closest hit's triangle loop, lean form (triangleTest stays for the any-hit loops and as the fallback). Same
// operations on the same values as triangleTest for every lane whose result is used; what changes:
//   * The reciprocal's range guard moves out of the loop: |det| = |e1 . (d x e2)| <= |e1| |e2| |d| (1 + 4 ulp); the host
//     bounds |e1| |e2| <= 2^100 (SceneLayout::triDetBounded) and the caller tests |d|^2 < 2^30 once per query, so
//     |det| < 2^126; below, results with |det| <= 1e-7 are discarded (Primitives.h:41) — exactly the operand range on which
//     ptm::rcp's fast path is proven equal to 1.0f / x. Queries that fail the test take the guarded loop.
//   * `b0 < 0 || b1 < 0 || b2 < 0` is decided as min3(b0, b1, b2) < 0: v_min3_f32 passes over NaN operands (a NaN weight
//     fails `< 0` in the reference too) and returns NaN only when all three are NaN (again no rejection); -0 is not < 0
//     either way.
//   * Only (distance, index, w1, w2) of the best hit travel through the loop, merged with selects (no exec-masked accept
//     block); w0 = 1 - (w1 + w2) is recomputed from the kept pair by the caller — the same operation on the same values.
//   * triangleTest's ONE wave-uniform exit (after the distance test) stays: tiles of the early bounces are coherent —
//     neighbouring pixels — and then whole waves do reject a triangle early. (No exit at all is the faster loop on
//     incoherent rays, tools/microbench/loops.hip: 157 -> 138 SIMD-cycles per triangle per wave, and the slower kernel:
//     same-box A/B -1.6 %.)
//   * EDGE CLASSES (kC1, kC2; pttri.h). A triangle whose edges run along coordinate axes (every wall and light panel of the
//     presets but two) loses the products with the exact zeros: 13 instead of 25 operations up to the distance test with two
//     such edges, 19 with one. Every lane of the wave tests the SAME triangle, so the body could be chosen per triangle
//     without divergence — but a scalar branch tree per triangle (35 scalar instructions, 9 branches) cost more than the
//     shorter bodies saved (same-box A/B -2.5 %: scalar instructions are not free beside vector ones,
//     tools/microbench/vgpr_banks.hip). So the host stores the triangles GROUPED BY CLASS (SceneLayout::triClassed /
//     triClassPack) and the loop becomes one loop per class: no dispatch at all. The visiting order is then no longer the
//     caller's, which matters where the reference's sequential rule `dist <= distance` (Primitives.h:52) decides between two
//     triangles hit at exactly the same distance: it ends on the HIGHEST index among them. kKeyed keeps (distance,
//     0xFFFFFFFE - original index) as one 64-bit key — distances that pass `dist > 0` order like their bit patterns — and
//     accepts a hit iff its key is SMALLER than the kept one: minimum distance, then highest original index; the initial key
//     (sphere distance, 0xFFFFFFFF) lets a triangle at exactly the sphere's distance win, as `<=` does. One v_cmp_lt_u64 in
//     place of one v_cmp_ngt_f32: the same issue cost. Exactness of the class forms, preconditions and the one case the
//     caller re-evaluates (a kept weight of exactly zero): pttri.h.
struct TriBest {
    float dist;    // the running `distance` (Primitives.h:52), shared with the sphere phase
    uint32_t key;  // 0xFFFFFFFF: no triangle accepted; kKeyed: 0xFFFFFFFE - original index; else the triangle's index
    float w1, w2;
};
constexpr uint32_t kNoTriangle = 0xffffffffu;
template <bool kPrimary, int kC1, int kC2, bool kKeyed>
__device__ __forceinline__ void triangleClassed(const float4* rows /* {v0, mat}, {e1, key}, {e2} */, const float4* prim /* {s, e2 . r}, {r} */,
                                                uint32_t index, vec3 o, vec3 d, unsigned long long liveMask, TriBest& best) {
    vec3 v0 = v3(0, 0, 0), ps = v3(0, 0, 0), pr = v3(0, 0, 0);
    float pe2r = 0;
    if constexpr (kPrimary) {   // the camera-origin test never looks at v0
        const float4 a = prim[0];
        ps = xyz(a);
        pe2r = a.w;
        pr = xyz(loadRow16(prim + 1));
    } else {
        v0 = xyz(loadRow16(rows));
    }
    const float4 rowE1 = rows[1];
    const pttri::Head h = pttri::head<kC1, kC2, kPrimary>(v0, xyz(rowE1), xyz(loadRow16(rows + 2)), ps, pr, pe2r, o, d);
    const uint32_t key = kKeyed ? asU(rowE1.w) : index;
    unsigned long long passMask = liveMask & maskOf(!(ptm::abs(h.det) <= 1e-7f)) & maskOf(!(h.dist <= 0.0f));
    if constexpr (kKeyed) {
        const unsigned long long mine = ((unsigned long long)asU(h.dist) << 32) | key, kept = ((unsigned long long)asU(best.dist) << 32) | best.key;
        passMask &= maskOf(mine < kept);
    } else {
        passMask &= maskOf(!(h.dist > best.dist));
    }
    if (passMask != 0ull) {
        float b0, b1, b2;
        pttri::weights<kC1, kC2>(h, d, b0, b1, b2);
        const unsigned long long hitMask = passMask & maskOf(!(__builtin_fminf(__builtin_fminf(b0, b1), b2) < 0));
        const bool hit = __builtin_amdgcn_inverse_ballot_w64(hitMask);
        best.dist = hit ? h.dist : best.dist;
        best.key = hit ? key : best.key;
        best.w1 = hit ? b1 : best.w1;
        best.w2 = hit ? b2 : best.w2;
    }
}
// One loop per edge class over the triangles stored for it; BODY(c1, c2, t) tests stored triangle t. The 17 class bounds travel
// as bytes in five scalar registers (SceneLayout::triClassPack) and every loop header extracts its two with s_bfe: as seventeen
// kernel-argument words the compiler evaluated all thirteen "is this class empty" conditions once per kernel, kept them as lane
// masks, spilled those to VGPR lanes and read them back with two v_readlane per loop header — 26 per query. The empty asm
// statements make the packed words opaque at each header, so that nothing about them is hoisted or kept.
struct ClassBounds {
    uint32_t w[5];
};
__device__ __forceinline__ ClassBounds classBounds(const SceneLayout& L) {
    return ClassBounds{{L.triClassPack[0], L.triClassPack[1], L.triClassPack[2], L.triClassPack[3], L.triClassPack[4]}};
}
template <int kCode>
__device__ __forceinline__ int classBegin(ClassBounds& b) {
    asm volatile("" : "+s"(b.w[kCode / 4]));
    return (int)((b.w[kCode / 4] >> (8 * (kCode % 4))) & 255u);
}
#define PTSS_FOR_TRIANGLES_BY_CLASS(L, BODY)                                                                        \
    do {                                                                                                            \
        ClassBounds _cb = classBounds(L);                                                                           \
        PTSS_TRI_CLASS_LOOP(_cb, 0, 0, BODY) PTSS_TRI_CLASS_LOOP(_cb, 0, 1, BODY) PTSS_TRI_CLASS_LOOP(_cb, 0, 2, BODY) PTSS_TRI_CLASS_LOOP(_cb, 0, 3, BODY) \
        PTSS_TRI_CLASS_LOOP(_cb, 1, 0, BODY) PTSS_TRI_CLASS_LOOP(_cb, 1, 2, BODY) PTSS_TRI_CLASS_LOOP(_cb, 1, 3, BODY)      \
        PTSS_TRI_CLASS_LOOP(_cb, 2, 0, BODY) PTSS_TRI_CLASS_LOOP(_cb, 2, 1, BODY) PTSS_TRI_CLASS_LOOP(_cb, 2, 3, BODY)      \
        PTSS_TRI_CLASS_LOOP(_cb, 3, 0, BODY) PTSS_TRI_CLASS_LOOP(_cb, 3, 1, BODY) PTSS_TRI_CLASS_LOOP(_cb, 3, 2, BODY)      \
    } while (0)
#define PTSS_TRI_CLASS_LOOP(cb, c1, c2, BODY) \
    for (int t = classBegin<(c1) * 4 + (c2)>(cb), tEnd = classBegin<(c1) * 4 + (c2) + 1>(cb); t < tEnd; ++t) { BODY(c1, c2, t) }

// the any-hit form of the same bodies (lineOfSight is an OR over independent tests: any order)
template <int kC1, int kC2>
__device__ __forceinline__ void triangleClassedAny(const float4* rows, vec3 o, vec3 d, float limit, unsigned long long& need, unsigned long long& blocked) {
    const pttri::Head h = pttri::head<kC1, kC2, false>(xyz(loadRow16(rows)), xyz(loadRow16(rows + 1)), xyz(loadRow16(rows + 2)), v3(0, 0, 0), v3(0, 0, 0), 0.0f, o, d);
    const unsigned long long passMask = need & maskOf(!(ptm::abs(h.det) <= 1e-7f)) & maskOf(!(h.dist <= 0.0f)) & maskOf(!(h.dist > limit));
    if (passMask != 0ull) {
        float b0, b1, b2;
        pttri::weights<kC1, kC2>(h, d, b0, b1, b2);
        const unsigned long long hitMask = passMask & maskOf(!(__builtin_fminf(__builtin_fminf(b0, b1), b2) < 0));
        blocked |= hitMask;
        need &= ~hitMask;
    }
}
// ... and for the TWO segments of a surface point (pairAnyHit): the origin part once, a direction part per segment
template <int kC1, int kC2>
__device__ __forceinline__ void triangleClassedPair(const float4* rows, vec3 o, vec3 dA, float limitA, vec3 dB, float limitB, unsigned long long& needA,
                                                    unsigned long long& needB, unsigned long long& blockedA, unsigned long long& blockedB) {
    const vec3 e1 = xyz(loadRow16(rows + 1)), e2 = xyz(loadRow16(rows + 2));
    const pttri::OriginPart p = pttri::originPart<kC1, kC2>(xyz(loadRow16(rows)), e1, e2, o);
    if (needA != 0ull) {
        const pttri::Head h = pttri::headFrom<kC1, kC2>(p, e1, e2, dA);
        const unsigned long long pass = needA & maskOf(!(ptm::abs(h.det) <= 1e-7f)) & maskOf(!(h.dist <= 0.0f)) & maskOf(!(h.dist > limitA));
        if (pass != 0ull) {
            float b0, b1, b2;
            pttri::weights<kC1, kC2>(h, dA, b0, b1, b2);
            const unsigned long long hit = pass & maskOf(!(__builtin_fminf(__builtin_fminf(b0, b1), b2) < 0));
            blockedA |= hit;
            needA &= ~hit;
        }
    }
    if (needB != 0ull) {
        const pttri::Head h = pttri::headFrom<kC1, kC2>(p, e1, e2, dB);
        const unsigned long long pass = needB & maskOf(!(ptm::abs(h.det) <= 1e-7f)) & maskOf(!(h.dist <= 0.0f)) & maskOf(!(h.dist > limitB));
        if (pass != 0ull) {
            float b0, b1, b2;
            pttri::weights<kC1, kC2>(h, dB, b0, b1, b2);
            const unsigned long long hit = pass & maskOf(!(__builtin_fminf(__builtin_fminf(b0, b1), b2) < 0));
            blockedB |= hit;
            needB &= ~hit;
        }
    }
}
// what the class bodies need of a query (pttri.h): a finite direction short enough to bound |det|, a finite origin
__device__ __forceinline__ bool classedQueryOk(vec3 o, vec3 d) { return waveAll(dot(d, d) < 0x1p30f) && waveAll(dot(o, o) < 0x1p100f); }
// ---- Primary (bounce 0) variants. Every eye ray starts at camera.position, so whatever the tests
// compute from the ORIGIN and the primitive alone is the same for all lanes and all pixels of a frame:
//   sphere:   v = o - centre,  c = dot(v,v) - r^2                    (Primitives.h:109,113)
//   triangle: s = o - v0,  r = cross(s, e1),  dot(e2, r)             (Primitives.h:46-49)
// primaryPrepKernel evaluates these once per camera with the very same operations; the per-lane work
// that is left is identical to the generic tests (same values, same order), minus 8 of 15 / 12 of 55
// instructions.

__device__ __forceinline__ bool sphereTestPrimary(float4 pv, vec3 d, float limit, float& t) {
    const float b = dot(d, xyz(pv)) * 2;
    float disc = (b * b) - 4 * pv.w;
    if (disc < 0) return false;
    disc = ptm::sqrt(disc);
    float t0 = (-b + disc) * 0.5f;
    float t1 = (-b - disc) * 0.5f;
    if (t0 < 0 && t1 < 0) return false;
    if (t0 > t1) {
        const float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    const float cand = (t0 < 0) ? t1 : t0;
    if (cand > limit) return false;
    t = cand;
    return true;
}

__device__ __forceinline__ TriHit triangleTestPrimary(const TriRows& tr, float4 ps /* s, dot(e2,r) */, float4 pr /* r */,
                                                      vec3 d, float limit, unsigned long long liveMask) {
    const vec3 e1 = xyz(tr.b), e2 = xyz(tr.c);
    const vec3 q = cross(d, e2);
    const float det = dot(e1, q);
    const float inverseDet = triRcp(det);  // 1 / det, Primitives.h:44; unused when |det| <= 1e-7
    const float dist = ps.w * inverseDet;
    const unsigned long long passMask = liveMask & maskOf(!(ptm::abs(det) <= 1e-7f)) & maskOf(!(dist <= 0.0f)) & maskOf(!(dist > limit));
    TriHit h;
    h.hit = false;
    h.hitMask = 0ull;
    h.dist = dist;
    h.w0 = h.w1 = h.w2 = 0;
    if (passMask != 0ull) {
        const float b1 = dot(xyz(ps), q) * inverseDet;
        const float b2 = dot(d, xyz(pr)) * inverseDet;
        const float b0 = 1.0f - (b1 + b2);
        h.hitMask = passMask & maskOf(!(b0 < 0)) & maskOf(!(b1 < 0)) & maskOf(!(b2 < 0));
        h.hit = __builtin_amdgcn_inverse_ballot_w64(h.hitMask);
        h.w0 = b0;
        h.w1 = b1;
        h.w2 = b2;
    }
    return h;
}

// ---- sphere candidate masks, CudaTracer.cu:127-133 / :438-444 through Primitives.h:107-118 -------------------------
// bit j of the result = "sphere j of this block of up to 32 passes the reference's discriminant test" — the very
// operations of the test above (Primitives.h:109-118), four spheres per trip: the four rows are fetched with one address and immediate offsets
// (the host pads the sphere rows to a multiple of four, packScene; a padding row's bit is dropped by the caller's `keep`
// mask), and each verdict enters the mask through the carry of one add (mask = 2 * mask + verdict: v_cmp + v_addc
// instead of v_cmp + v_cndmask + v_or and a v_mov for the bit). That leaves the first sphere in the highest bit; one
// v_bfrev + shift puts sphere j at bit j, which the candidate loops need (they walk in index order).
// rev = 2 * rev + !(disc < 0) for one sphere, disc = b * b - 4 * c (Primitives.h:115-118). `disc < 0` is decided as
// `b * b < 4 * c`: a correctly rounded difference of two floats is negative exactly when the first is the smaller
// (gradual underflow: it is zero only for equal operands; inf - inf = NaN and a NaN operand make both forms false) —
// hipcc performs the same fold on its own. The verdict goes from VCC into the mask as the carry of one add.
__device__ __forceinline__ void shiftInMayHit(uint32_t& rev, float bb, float c4) {
    asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rev) : "v"(bb), "v"(c4) : "vcc");
}
// kBounded: the same verdict from two instructions less, on bounded geometry (SceneLayout::sphereBounded, set by
// ptss_create: every |coordinate| <= 1e15 and every sphere radius in [1e-12, 1e15]; the camera is checked per frame; ray
// origins — the camera or points on primitives — are then bounded as well). With h = d.v the reference compares
// RN((2h)^2) with 4c; doubling and quadrupling are exact, so that is 4 RN(h^2) < 4c, i.e. RN(h^2) < c, unless (a) 4c
// overflows — c < 2^105 here —, (b) 4 h^2 overflows — then h^2 >= 2^126 > c and both forms say "may hit" —, or (c) h^2 is
// subnormal and loses bits that 4 h^2 keeps — then h^2 < 2^-126, while c is zero or at least an ulp of r^2 >= 1e-24 in
// magnitude (a difference of two floats), so its sign decides both forms alike (c = 0: neither `<` holds). NaN or
// infinite operands make both compares false. Pinned on adversarial operands by tests/test_sphere_forms.py.
// kCull (the closest hit's masks and the many-sphere visits) also drops the spheres BEHIND the origin, which the reference rejects two lines further down (both roots negative,
// Primitives.h:126-127) — the sphere a reflected ray has just left above all (origin bumped 1e-4 off it: c ~ 2e-4 r, h ~ r), a
// candidate of every such ray otherwise, and every sphere the ray's line meets behind it. The mask compares h * m with c,
// m = min(h * 2^-18, h): h for h <= 0 — the very product h * h, nothing changes ahead of the origin — and 2^-18 h for h > 0.
// A sphere dropped that way has h > 0 and c > 2^-18 RN(h^2) =: k H. Then the reference computes disc = 4 RN(H - c) <
// 4 H (1 - k)(1 + 2^-24), s = RN(sqrt(disc)) < 2 h (1 + 2^-25)(1 - 2^-19 + 2^-24)(1 + 2^-24) < 2 h = b (exact doubling), so
// t0 = RN(-b + s) / 2 < 0 and t1 = RN(-b - s) / 2 < 0: rejected whatever the running distance — or disc < 0 and it was
// rejected before. (Scaling by 2^-18 is exact; where it underflows, floats are 2^-149 apart and c > RN(k H) still means
// c > k H. A NaN h stays a NaN m: kept, as before.) Pinned on corner operands, random bit patterns and operands a few ulps
// around the threshold by tests/test_sphere_behind.py. Same-box A/B: c3 +1.2 ... +2.0 %, c5 +1.7 %, c2 +0.6 %; in the 38-primitive
// scenes' shadow passes as well it bought nothing more (a blocked segment leaves at its first hit): they keep the plain mask.
// h for h <= 0, h * 2^-18 for h > 0 (one multiply, one v_min_f32)
template <bool kCull>
__device__ __forceinline__ float aheadFactor(float h) {
    if constexpr (kCull) {
        const float hk = h * 0x1p-18f;
        float m;
        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(hk), "v"(h));
        return m;
    } else {
        return h;
    }
}
template <bool kBounded, bool kCull = false>
__device__ __forceinline__ void shiftInSphere(uint32_t& rev, float4 sp, vec3 o, vec3 d) {  // Primitives.h:109-118
    const vec3 v = o - xyz(sp);
    if constexpr (kBounded) {
        const float h = dot(d, v);
        const float c = dot(v, v) - sp.w;
        shiftInMayHit(rev, h * aheadFactor<kCull>(h), c);
    } else {
        const float b = dot(d, v) * 2;
        const float c = dot(v, v) - sp.w;
        shiftInMayHit(rev, b * b, 4 * c);
    }
}
template <bool kBounded>
__device__ __forceinline__ void shiftInSpherePrimary(uint32_t& rev, float4 pv, vec3 d) {  // the same from the camera-origin precomputes
    if constexpr (kBounded) {
        const float h = dot(d, xyz(pv));
        shiftInMayHit(rev, h * h, pv.w);
    } else {
        const float b = dot(d, xyz(pv)) * 2;
        shiftInMayHit(rev, b * b, 4 * pv.w);
    }
}
template <bool kPrimary, bool kBounded>
__device__ __forceinline__ uint32_t sphereCandidates(const float4* rows, int cnt, vec3 o, vec3 d) {
    const int trips = (cnt + 3) >> 2;  // wave-uniform, 1..8
    uint32_t rev = 0;
    for (int g = 0; g < trips; ++g) {
        const float4 r0 = rows[4 * g], r1 = rows[4 * g + 1], r2 = rows[4 * g + 2], r3 = rows[4 * g + 3];
        if constexpr (kPrimary) {
            shiftInSpherePrimary<kBounded>(rev, r0, d);
            shiftInSpherePrimary<kBounded>(rev, r1, d);
            shiftInSpherePrimary<kBounded>(rev, r2, d);
            shiftInSpherePrimary<kBounded>(rev, r3, d);
        } else {
            shiftInSphere<kBounded, true>(rev, r0, o, d);
            shiftInSphere<kBounded, true>(rev, r1, o, d);
            shiftInSphere<kBounded, true>(rev, r2, o, d);
            shiftInSphere<kBounded, true>(rev, r3, o, d);
        }
    }
    return __builtin_bitreverse32(rev) >> (32 - 4 * trips);
}
// two spheres per trip: for the shadow passes, where the registers are needed elsewhere (four rows in flight there
// push the 72-VGPR kernel into scratch)
template <bool kBounded>
__device__ __forceinline__ uint32_t sphereCandidatesPairs(const float4* rows, int cnt, vec3 o, vec3 d) {
    const int trips = (cnt + 1) >> 1;  // 1..16
    uint32_t rev = 0;
    for (int g = 0; g < trips; ++g) {
        const float4 r0 = rows[2 * g], r1 = rows[2 * g + 1];
        shiftInSphere<kBounded>(rev, r0, o, d);
        shiftInSphere<kBounded>(rev, r1, o, d);
    }
    return __builtin_bitreverse32(rev) >> (32 - 2 * trips);
}
template <bool kBounded>
__device__ __forceinline__ uint32_t sphereCandidatesStridedPairs(const float4* first, int stride, int cnt, vec3 o, vec3 d) {
    const int trips = (cnt + 1) >> 1;
    uint32_t rev = 0;
    for (int g = 0; g < trips; ++g) {
        const float4* p = first + 2 * g * stride;
        const float4 r0 = p[0], r1 = p[stride];
        shiftInSphere<kBounded>(rev, r0, o, d);
        shiftInSphere<kBounded>(rev, r1, o, d);
    }
    return __builtin_bitreverse32(rev) >> (32 - 2 * trips);
}
__device__ __forceinline__ uint32_t lowBits(int cnt) { return (cnt >= 32) ? 0xffffffffu : ((1u << cnt) - 1u); }
__device__ __forceinline__ uint32_t lowBitsClamped(int cnt) { return (cnt <= 0) ? 0u : lowBits(cnt); }

// ---- closest hit over spheres then triangles, CudaTracer.cu:121-141 ---------------------------
// Spheres, 32 at a time: a uniform pass records in a per-lane bit mask which spheres survive the
// discriminant test; then every lane resolves ITS OWN candidates in index order. A sphere that
// fails the discriminant test never changes `distance`, so visiting only the candidates, in the
// same order, accepts exactly what the reference's full loop accepts — but the square-root path
// runs a few times per lane instead of once per sphere for the whole wave.
struct Hit {
    float distance;
    int kind, idx;  // kind: 0 none, 1 sphere, 2 triangle
    float w0, w1, w2;
};

// ---- Scenes with many spheres (SceneLayout::accelSpheres; derivation of the test and of its constants: packScene in
// ptss_api.hip). The spheres sit in spatially sorted chunks of kChunkSpheres with a bounding sphere each. chunkMask is
// the wave-uniform pass over 32 chunk bounds: bit k = "this lane's ray may touch chunk k" — a conservative test that
// only ever skips spheres whose reference discriminant is certainly negative. Each lane then walks ITS chunks (per-lane
// gathers) with the reference's own tests. The visiting order is no longer the reference's, which matters only when two
// spheres are hit at exactly the same distance: the sequential `<=` rule ends on the HIGHEST index among them, so the
// closest hit keeps (minimum distance, highest original index) — identical for the finite distances this mode is
// restricted to.
constexpr int kQueueCapConst = 2 * 64;  // = kQueueCap (static_assert below): segments per wave queue plane
constexpr float kAccelMu = 5e-3f + 5e-3f * 5e-3f;   // m + m^2
constexpr float kAccelDirEps = 1e-5f;               // | |d|^2 - 1 | up to which a direction counts as unit
constexpr float kAccelQ = 0.25f * (1.0f + 2e-5f) / (1.0f - kAccelMu) * (1.0f + 1e-6f);   // (1 + 2 eps) / (4 (1 - mu)), rounded up

// One chunk bound's verdict ("this lane's ray may touch the chunk") shifted into `rev` through the carry, as shiftInSphere does
// for spheres. ONE test (derivation: packScene): with t = dv - |dv| = 2 min(dv, 0) — exact, no compare, no select —
// vv - kAccelQ t^2 is (a lower bound of) the squared distance of the chunk's centre from the RAY, the half line t >= 0: the
// line's distance while the closest approach lies ahead of the origin, the origin's own distance once it lies behind. The
// chunk is skipped when that exceeds the stored bound; the compare's wave mask is handed to v_addc as its carry-in SGPR
// pair: no v_cndmask, no v_or, no v_mov for the bit. (Until round 3 the line and a separate "wholly behind the origin's
// plane" test — two more compares, a multiply, an fma and three scalar instructions per bound, and a looser verdict: a ray
// leaving a chunk it starts beside was still sent into it.)
__device__ __forceinline__ void shiftInChunk(uint32_t& rev, float4 b, vec3 o, vec3 d) {
    const vec3 v = o - xyz(b);
    const float dv = dot(d, v);
    const float vv = dot(v, v);
    const float t = dv - ptm::abs(dv);
    const unsigned long long may = ~maskOf(ptm::fma(-kAccelQ, t * t, vv) > b.w);   // not provably out of reach (a NaN lands here too)
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(rev) : "s"(may) : "vcc");
}
// The same for a ray that starts at the camera (bounce 0): the row holds v = o - C and vv - bound, evaluated once per camera by
// primaryPrepKernel with the very same subtraction (the difference rounded DOWN: it can only keep a chunk) — 8 instructions
// instead of 14 per bound.
__device__ __forceinline__ void shiftInChunkPrimary(uint32_t& rev, float4 pv, vec3 d) {
    const float dv = dot(d, xyz(pv));
    const float t = dv - ptm::abs(dv);
    const unsigned long long may = ~maskOf(ptm::fma(-kAccelQ, t * t, pv.w) > 0.0f);
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(rev) : "s"(may) : "vcc");
}
// four bounds per trip (one address, immediate offsets; the host pads the bound rows to a multiple of four and the padding's
// bits are dropped here)
template <bool kPrimary>
__device__ __forceinline__ uint32_t chunkMask(const float4* bounds, int cnt, vec3 o, vec3 d, bool unitDir) {
    const int trips = (cnt + 3) >> 2;  // wave-uniform, 1..8
    uint32_t rev = 0;
    for (int g = 0; g < trips; ++g) {
        const float4 b0 = bounds[4 * g], b1 = bounds[4 * g + 1], b2 = bounds[4 * g + 2], b3 = bounds[4 * g + 3];
        if constexpr (kPrimary) {
            shiftInChunkPrimary(rev, b0, d);
            shiftInChunkPrimary(rev, b1, d);
            shiftInChunkPrimary(rev, b2, d);
            shiftInChunkPrimary(rev, b3, d);
        } else {
            shiftInChunk(rev, b0, o, d);
            shiftInChunk(rev, b1, o, d);
            shiftInChunk(rev, b2, o, d);
            shiftInChunk(rev, b3, o, d);
        }
    }
    const uint32_t all = (cnt >= 32) ? 0xffffffffu : ((1u << cnt) - 1u);
    return unitDir ? ((__builtin_bitreverse32(rev) >> (32 - 4 * trips)) & all) : all;
}

#include "ptss_diag.h"

// Candidate mask of ONE chunk for a lane that gathers its own rows (lanes sit in different chunks): visit i reads slot
// i ^ (chunk mod kChunkSpheres), so that the 16-byte gathers of a wave spread over the LDS banks; verdicts enter through
// the carry (shiftInSphere), so visit i lands in bit kChunkSpheres - 1 - i. chunkSlot() turns a bit of that mask back into
// the sphere's slot inside the chunk. The traversal is order-free (ties go by original index). Where the image is staged in
// LDS and the sphere rows start on a 256-byte boundary (they do: packScene puts them first, the dynamic LDS is aligned), a
// row's address is (chunk's address ^ (chunk mod 16) << 4) ^ (i << 4): ONE v_xor with a constant per row instead of add, and,
// shift-add (round 3; -2 of 16 instructions per sphere).
typedef __attribute__((address_space(3))) const float4 LdsRow;
__device__ __forceinline__ uint32_t chunkCandidates(const float4* spheres /* sc + L.offSphere */, int base, int chunk, vec3 o, vec3 d) {
    static_assert(kChunkSpheres * sizeof(float4) <= 256, "a chunk's rows must not straddle the 256-byte window the XOR walks");
    uint32_t rev = 0;
    const int twist = chunk & (kChunkSpheres - 1);
#if __HIP_DEVICE_COMPILE__   // (the host pass of this file only parses device functions; it has no LDS address space)
    if (__builtin_amdgcn_is_shared(spheres)) {   // decided at compile time wherever the image's address space is known
        const uint32_t first = (uint32_t)(uintptr_t)(LdsRow*)spheres;
        if ((first & 255u) == 0u) {   // wave-uniform
            uint32_t x = (first + (uint32_t)base * (uint32_t)sizeof(float4)) ^ ((uint32_t)twist << 4);
            asm volatile("" : "+v"(x));   // keep it one value: the compiler would re-associate it into (i ^ twist) << 4 ^ base per row
#pragma unroll
            for (int i = 0; i < kChunkSpheres; ++i) shiftInSphere<true, true>(rev, *(LdsRow*)(uintptr_t)(x ^ ((uint32_t)i << 4)), o, d);
            return rev;
        }
    }
#endif
#pragma unroll 4
    for (int i = 0; i < kChunkSpheres; ++i) shiftInSphere<true, true>(rev, spheres[base + (i ^ twist)], o, d);
    return rev;
}
__device__ __forceinline__ int chunkSlot(int bit, int chunk) { return ((kChunkSpheres - 1 - bit) ^ chunk) & (kChunkSpheres - 1); }

// The chunk bits of up to 128 chunks (4 words) are gathered first and walked in ONE per-lane loop: the wave then runs as
// long as its busiest lane's TOTAL, not the sum over 32-chunk groups of each group's busiest lane.
struct ChunkBits {
    uint32_t w[4];
};
template <bool kPrimary = false>
__device__ __forceinline__ ChunkBits chunkBits128(const float4* sc, const SceneLayout& L, int g0, vec3 o, vec3 d, bool unitDir, bool live) {
    ChunkBits b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = g0 + 32 * q;
        const int left = L.numChunks - g;  // wave-uniform
        b.w[q] = (left > 0) ? chunkMask<kPrimary>(sc + (kPrimary ? L.offPrimChunk : L.offChunk) + g, left < 32 ? left : 32, o, d, unitDir) : 0u;
        if (!live) b.w[q] = 0u;
    }
    return b;
}
__device__ __forceinline__ bool anyChunk(const ChunkBits& b) { return (b.w[0] | b.w[1] | b.w[2] | b.w[3]) != 0u; }
__device__ __forceinline__ int popChunk(ChunkBits& b) {  // lowest set bit, removed
    const int q = b.w[0] ? 0 : (b.w[1] ? 1 : (b.w[2] ? 2 : 3));
    const uint32_t word = q == 0 ? b.w[0] : (q == 1 ? b.w[1] : (q == 2 ? b.w[2] : b.w[3]));
    const int k = __builtin_ctz(word);
    const uint32_t rest = word & (word - 1u);
    b.w[0] = q == 0 ? rest : b.w[0];
    b.w[1] = q == 1 ? rest : b.w[1];
    b.w[2] = q == 2 ? rest : b.w[2];
    b.w[3] = q == 3 ? rest : b.w[3];
    return 32 * q + k;
}


__device__ __forceinline__ bool anySphereChunked(const float4* sc, const SceneLayout& L, vec3 lo, vec3 w_i, float distance, bool live) {
    const bool unitDir = ptm::abs(dot(w_i, w_i) - 1.0f) <= kAccelDirEps;
    bool occluded = false;
    for (int g0 = 0; g0 < L.numChunks; g0 += 128) {
        ChunkBits chunks = chunkBits128(sc, L, g0, lo, w_i, unitDir, live && !occluded);
        while (anyChunk(chunks)) {
            const int chunk = g0 + popChunk(chunks);
            const int base = chunk * kChunkSpheres;
            uint32_t mask = chunkCandidates(sc + L.offSphere, base, chunk, lo, w_i);
            while (mask != 0) {
                const int j = chunkSlot(__builtin_ctz(mask), chunk);
                mask &= mask - 1;
                float t;
                if (sphereTest(sc[L.offSphere + base + j], lo, w_i, distance, t)) {
                    occluded = true;
                    mask = 0;
                    chunks.w[0] = chunks.w[1] = chunks.w[2] = chunks.w[3] = 0u;
                }
            }
        }
    }
    return occluded;
}

// ---- The same traversal with the work REGROUPED across the wave. In a dense scene an incoherent ray touches 30-50 chunks
// and the counts differ widely between lanes: walking them lane by lane keeps 34 % of the lanes busy
// (tools/stress_counters.sh). Here every lane publishes its ray in the wave's LDS area, an exclusive scan of the chunk
// counts numbers all (ray, chunk) pairs of the wave, every lane writes its pairs into a list at its scan position, and
// each pass hands 64 consecutive pairs to the 64 lanes: lane l reads pair q = (owner, chunk), tests the chunk's
// spheres against the OWNER's ray, and folds what it finds into the owner's slot with one 64-bit LDS minimum on the key
// (the shadow passes' regrouped part, anySpheresHybrid, still FINDS pair q: owner by bisection over the scan, chunk as the
// owner's r-th set bit — its tables sit in strided half planes of the segment queue)
// (distance bits, ~original index): minimum distance first, highest original index among equals — the order-free form
// of the reference's sequential rule (distances are >= 0 here, so their bit patterns order like the values; -0 counts as
// +0, all-NaN rays tie on the distance and end on the highest index, as the sequential loop does). The owner finally
// recomputes the winner's distance with the reference's own test, so the value it keeps has the reference's bits.
// (Shadow rays keep the per-lane walk: most of them are blocked within their first chunks, and that early exit beats
// balanced scheduling — the regrouped any-hit measured 15.2 against 11.0 ms per pass on the configs[5] scene.)
__device__ __forceinline__ uint32_t nthSetBit(uint32_t word, uint32_t r) {  // position of the r-th (0-based) set bit
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t width = 16; width >= 1; width >>= 1) {
        const uint32_t low = (uint32_t)__builtin_popcount(word & ((1u << width) - 1u));
        const bool up = r >= low;
        r -= up ? low : 0u;
        word = up ? (word >> width) : word;
        pos += up ? width : 0u;
    }
    return pos;
}

constexpr uint32_t kPairCap = 2 * 5 * 64;   // 16-bit words in the five 64-word tables between the rays and the keys
constexpr uint32_t kCandCap = 8 * kQueueCapConst + kQueueCapConst / 4 - 13 * 64;   // what the wave's LDS area holds behind the tables: 224 words
static_assert(kCandCap >= 128, "the candidate queue must take a full trip after a drain");

template <bool kPrimary>
__device__ __forceinline__ void closestSpheresRegrouped(const float4* sc, const float4* cold, const SceneLayout& L, vec3 o, vec3 d,
                                                        bool live, Hit& h, uint32_t* ws) {
    const uint32_t lane = __lane_id();
    float* rayTab = reinterpret_cast<float*>(ws);                                    // [6][64]
    uint16_t* pairQ = reinterpret_cast<uint16_t*>(ws + 6 * 64);                      // [kPairCap]: owner lane | chunk (of this group of 128) << 6
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + 11 * 64);  // [64]
    uint32_t* candQ = ws + 13 * 64;                                                  // [kCandCap]: owner lane | sorted sphere position << 8
    const int* orig = reinterpret_cast<const int*>(cold + L.offSphereOrig);  // global memory (SceneLayout::ldsVec4)
    const int* posOf = reinterpret_cast<const int*>(cold + L.offSpherePos);
    const bool unitDir = ptm::abs(dot(d, d) - 1.0f) <= kAccelDirEps;
    // CANDIDATES, second regrouping (round 3). A (ray, chunk) pair finds few candidates among its 16 spheres — the line of a
    // ray that touches a chunk's bound meets 0.3 of the chunk's spheres on average — so resolving them where they are found
    // (a per-lane loop inside every pass: as many trips as the busiest lane has candidates, a tenth of the lanes working) was
    // the largest single piece of a mid-bounce launch (ablation builds, profiles/README.md). Instead every pass only APPENDS
    // its candidates — (owner, sphere) words, ranked by ballot — to a queue in the wave's LDS area, and the queue is resolved
    // 64 at a time with every lane busy: square root, roots, key, one LDS minimum into the owner's slot. Any order is fine
    // (the merge is a minimum on (distance, ~original index)); the queue is drained whenever a trip might not fit, and at the
    // end. (The shadow passes' regrouped part keeps resolving in place: the same queue there measured +-0 — a blocked segment
    // leaves at its first hit, and most do.)
    uint32_t candCount = 0;   // wave-uniform
    auto resolveCandidates = [&](uint32_t n) {   // the last n <= 64 entries of the queue
        const bool have = lane < n;
        const uint32_t e = candQ[candCount - n + (have ? lane : 0u)];
        const uint32_t owner = e & 63u;
        const int pos = (int)(e >> 8);
        const vec3 ro = v3(rayTab[0 * 64 + owner], rayTab[1 * 64 + owner], rayTab[2 * 64 + owner]);
        const vec3 rd = v3(rayTab[3 * 64 + owner], rayTab[4 * 64 + owner], rayTab[5 * 64 + owner]);
        float t;
        if (have && sphereTest(sc[L.offSphere + pos], ro, rd, ptm::inf(), t)) {
            const uint32_t tb = (t != t) ? 0u : asU(t + 0.0f);
            atomicMin(&best[owner], ((unsigned long long)tb << 32) | (unsigned long long)(0xffffffffu - (uint32_t)orig[pos]));
        }
        candCount -= n;
    };
    rayTab[0 * 64 + lane] = o.x;
    rayTab[1 * 64 + lane] = o.y;
    rayTab[2 * 64 + lane] = o.z;
    rayTab[3 * 64 + lane] = d.x;
    rayTab[4 * 64 + lane] = d.y;
    rayTab[5 * 64 + lane] = d.z;
    best[lane] = ~0ull;
    for (int g0 = 0; g0 < L.numChunks; g0 += 128) {
        ChunkBits mine = chunkBits128<kPrimary>(sc, L, g0, o, d, unitDir, live);
        // PAIRS. Every lane writes its (owner lane, chunk) pairs — 16-bit words, ascending chunks — into the wave's pair list
        // at the position an exclusive scan of the counts gives it; a pass then reads one word per lane. (Until round 3 a pass
        // FOUND its pairs: bisection over the scan for the owner, the owner's four bit words, the r-th set bit — 120 vector
        // instructions and eleven dependent LDS round trips per pass; the expansion is one loop per 128 chunks with as many
        // trips as the busiest lane has chunks.) A list holds kPairCap pairs; what does not fit stays in the lanes' bits for
        // the next round.
        for (;;) {
            const uint32_t cnt = (uint32_t)(__builtin_popcount(mine.w[0]) + __builtin_popcount(mine.w[1]) + __builtin_popcount(mine.w[2]) +
                                            __builtin_popcount(mine.w[3]));
            uint32_t incl = cnt;  // inclusive scan over the lanes
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) {
                const uint32_t below = (uint32_t)__shfl_up((int)incl, off);
                incl += (lane >= off) ? below : 0u;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63);  // wave-uniform
            if (total == 0u) break;
            uint32_t pos = incl - cnt;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                while (waveAny(mine.w[w] != 0u && pos < kPairCap)) {
                    if (mine.w[w] != 0u && pos < kPairCap) {
                        const uint32_t k = (uint32_t)__builtin_ctz(mine.w[w]);
                        mine.w[w] &= mine.w[w] - 1u;
                        pairQ[pos++] = (uint16_t)(lane | ((32u * (uint32_t)w + k) << 6));
                    }
                }
            }
            const uint32_t n = total < kPairCap ? total : kPairCap;
            waveLdsFence();
            for (uint32_t q0 = 0; q0 < n; q0 += 64) {
                const bool work = q0 + lane < n;
                const uint32_t e = pairQ[work ? q0 + lane : 0u];
                const uint32_t owner = e & 63u;
                const int chunk = g0 + (int)(e >> 6);
                const int base = chunk * kChunkSpheres;
                const vec3 ro = v3(rayTab[0 * 64 + owner], rayTab[1 * 64 + owner], rayTab[2 * 64 + owner]);
                const vec3 rd = v3(rayTab[3 * 64 + owner], rayTab[4 * 64 + owner], rayTab[5 * 64 + owner]);
                uint32_t mask = chunkCandidates(sc + L.offSphere, base, chunk, ro, rd);
                if (!work) mask = 0;
                while (waveAny(mask != 0)) {   // one trip per candidate of the busiest lane: append, do not resolve
                    if (candCount + 64u > kCandCap) {   // wave-uniform: make room first
                        waveLdsFence();
                        while (candCount >= 64u) resolveCandidates(64u);
                        waveLdsFence();
                    }
                    const bool has = mask != 0;
                    const unsigned long long m = __ballot(has);
                    if (has) {
                        const int j = chunkSlot(__builtin_ctz(mask), chunk);
                        mask &= mask - 1;
                        candQ[candCount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = owner | ((uint32_t)(base + j) << 8);
                    }
                    candCount += (uint32_t)__popcll(m);
                }
            }
            waveLdsFence();
            if (total <= kPairCap) break;
        }
        waveLdsFence();
        while (candCount != 0u) resolveCandidates(candCount < 64u ? candCount : 64u);
        waveLdsFence();
    }
    const unsigned long long won = best[lane];
    PTSS_DIAG_CULL(sc, L, o, d, unitDir, live, won);
    if (live && won != ~0ull) {
        const int pos = posOf[0xffffffffu - (uint32_t)won];
        float t;
        (void)sphereTest(sc[L.offSphere + pos], o, d, ptm::inf(), t);  // the winner's distance, with the reference's bits
        h.distance = t;
        h.kind = 1;
        h.idx = pos;
    }
    waveLdsFence();
}

// ---- Shadow rays of a dense queue pass, hybrid: a blocked segment is usually blocked within its first chunks, so every
// lane walks up to kWarmChunks of its own chunks first (cheap early exits); what is left belongs to the long walkers —
// the segments that reach their light have to visit all 30-50 chunks — and is regrouped across the wave like the closest
// hit's work (closestSpheresRegrouped), the merge being "set the owner's blocked flag". The tables live in the half of
// the wave's queue planes that the current pass does not read (tab[plane] = that half of plane `plane`), which is why
// the caller uses this only for a pass whose other half is free.
// chunks of 16 in kd order, configs[4]'s scene at S = 4, same box: 0: 4,571, 1: 4,589, 2: 4,597-4,605, 3: 4,585, 4: 4,535,
// 8: 4,270, 16: 4,189 Mrays/s; again with round 3's tighter bounds: 0: 5,866, 1: 5,901, 2: 5,931, 3: 5,835, 4: 5,738
constexpr int kWarmChunks = 2;

__device__ __forceinline__ bool anySpheresHybrid(const float4* sc, const SceneLayout& L, const float* seg, float* tab, vec3 lo,
                                                 vec3 w_i, float distance, bool have) {
    const uint32_t lane = __lane_id();
    uint32_t* bits0 = reinterpret_cast<uint32_t*>(tab + 0 * kQueueCapConst);
    uint32_t* bits1 = reinterpret_cast<uint32_t*>(tab + 1 * kQueueCapConst);
    uint32_t* bits2 = reinterpret_cast<uint32_t*>(tab + 2 * kQueueCapConst);
    uint32_t* bits3 = reinterpret_cast<uint32_t*>(tab + 3 * kQueueCapConst);
    uint32_t* startTab = reinterpret_cast<uint32_t*>(tab + 4 * kQueueCapConst);
    uint32_t* blocked = reinterpret_cast<uint32_t*>(tab + 5 * kQueueCapConst);
    const bool unitDir = ptm::abs(dot(w_i, w_i) - 1.0f) <= kAccelDirEps;
    bool occluded = false;
    for (int g0 = 0; g0 < L.numChunks; g0 += 128) {
        ChunkBits mine = chunkBits128(sc, L, g0, lo, w_i, unitDir, have && !occluded);
        for (int it = 0; it < kWarmChunks; ++it) {  // own walk
            if (!waveAny(anyChunk(mine))) break;
            if (anyChunk(mine)) {
                const int chunk = g0 + popChunk(mine);
                const int base = chunk * kChunkSpheres;
                uint32_t mask = chunkCandidates(sc + L.offSphere, base, chunk, lo, w_i);
                while (mask != 0) {
                    const int j = chunkSlot(__builtin_ctz(mask), chunk);
                    mask &= mask - 1;
                    float t;
                    if (sphereTest(sc[L.offSphere + base + j], lo, w_i, distance, t)) {
                        occluded = true;
                        mask = 0;
                        mine.w[0] = mine.w[1] = mine.w[2] = mine.w[3] = 0u;
                    }
                }
            }
        }
        if (!waveAny(anyChunk(mine))) continue;
        // the rest, regrouped
        bits0[lane] = mine.w[0];
        bits1[lane] = mine.w[1];
        bits2[lane] = mine.w[2];
        bits3[lane] = mine.w[3];
        blocked[lane] = occluded ? 1u : 0u;
        const uint32_t cnt = (uint32_t)(__builtin_popcount(mine.w[0]) + __builtin_popcount(mine.w[1]) + __builtin_popcount(mine.w[2]) +
                                        __builtin_popcount(mine.w[3]));
        uint32_t incl = cnt;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t below = (uint32_t)__shfl_up((int)incl, off);
            incl += (lane >= off) ? below : 0u;
        }
        startTab[lane] = incl - cnt;
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        waveLdsFence();
        for (uint32_t q0 = 0; q0 < total; q0 += 64) {
            const uint32_t q = q0 + lane;
            const bool valid = q < total;
            uint32_t a = 0, b = 64;
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const uint32_t mid = (a + b) >> 1;
                const bool right = startTab[mid] <= q;
                a = right ? mid : a;
                b = right ? b : mid;
            }
            const uint32_t owner = valid ? a : lane;
            uint32_t r = valid ? q - startTab[owner] : 0u;
            uint32_t word = 0, wordIdx = 0;
            bool found = false;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t bits = (w == 0 ? bits0 : (w == 1 ? bits1 : (w == 2 ? bits2 : bits3)))[owner];
                const uint32_t pc = (uint32_t)__builtin_popcount(bits);
                const bool here = !found && r < pc;
                word = here ? bits : word;
                wordIdx = here ? (uint32_t)w : wordIdx;
                found = found || here;
                r -= (!found) ? pc : 0u;
            }
            const bool work = valid && found && blocked[owner] == 0u;
            const int chunk = g0 + (int)(32u * wordIdx + nthSetBit(word, r));
            const int base = (work ? chunk : 0) * kChunkSpheres;
            const vec3 so = v3(seg[0 * kQueueCapConst + owner], seg[1 * kQueueCapConst + owner], seg[2 * kQueueCapConst + owner]);
            const vec3 sd = v3(seg[3 * kQueueCapConst + owner], seg[4 * kQueueCapConst + owner], seg[5 * kQueueCapConst + owner]);
            const float reach = seg[6 * kQueueCapConst + owner];
            uint32_t mask = chunkCandidates(sc + L.offSphere, base, chunk, so, sd);
            if (!work) mask = 0;
            while (mask != 0) {
                const int j = chunkSlot(__builtin_ctz(mask), chunk);
                mask &= mask - 1;
                float t;
                if (sphereTest(sc[L.offSphere + base + j], so, sd, reach, t)) {
                    blocked[owner] = 1u;
                    mask = 0;
                }
            }
        }
        waveLdsFence();
        occluded = blocked[lane] != 0u;
        waveLdsFence();
    }
    return occluded;
}

// the triangle half of lineOfSight alone (the sphere half having been answered by anySpheresHybrid)
template <bool kPrimary, bool kAccel, bool kBounded>
__device__ __forceinline__ Hit closestHit(const float4* sc, const float4* cold, const SceneLayout& L, vec3 o, vec3 d, bool live, uint32_t* ws) {
    Hit h;
    h.distance = ptm::inf();
    h.kind = 0;
    h.idx = 0;
    h.w0 = h.w1 = h.w2 = 0;
    if constexpr (kAccel) closestSpheresRegrouped<kPrimary>(sc, cold, L, o, d, live, h, ws);
    for (int base = 0; base < (kAccel ? 0 : L.numSpheres); base += 32) {
        const int cnt = (L.numSpheres - base < 32) ? (L.numSpheres - base) : 32;
        uint32_t mask = sphereCandidates<kPrimary, kBounded>(sc + (kPrimary ? L.offPrimSphere : L.offSphere) + base, cnt, o, d);
        mask &= live ? lowBits(cnt) : 0u;
        PTSS_DIAG_CANDIDATES(mask, live, 0);
        while (mask != 0) {
            const int j = __builtin_ctz(mask);
            mask &= mask - 1;
            float t;
            const bool acc = kPrimary ? sphereTestPrimary(sc[L.offPrimSphere + base + j], d, h.distance, t)
                                      : sphereTest(sc[L.offSphere + base + j], o, d, h.distance, t);
            if (acc) {
                h.distance = t;
                h.kind = 1;
                h.idx = base + j;
            }
        }
    }
    const unsigned long long liveMask = maskOf(live);
    if (L.triClassed) {
        // The triangles are stored grouped by edge class. One test per query (not per triangle) admits the class bodies:
        // |d|^2 < 2^30 bounds |det| below the reciprocal's fast range, and with a finite origin every product the class
        // forms leave out is an exact zero (pttri.h). A wave that fails it (a NaN or enormous ray) walks the triangles in the
        // CALLER's order with the guarded general test below: the reference's sequential rule, NaNs included.
        if (classedQueryOk(o, d)) {
            TriBest best{h.distance, kNoTriangle, 0.0f, 0.0f};
#define PTSS_CLOSEST_BODY(c1, c2, t) \
    triangleClassed<kPrimary, c1, c2, true>(sc + L.offTri + 3 * t, sc + L.offPrimTri + 2 * t, 0u, o, d, liveMask, best);
            PTSS_FOR_TRIANGLES_BY_CLASS(L, PTSS_CLOSEST_BODY);
#undef PTSS_CLOSEST_BODY
            if (waveAny(best.key != kNoTriangle)) {
                const int* posOf = reinterpret_cast<const int*>(sc + L.offTriPos);
                int pos = 0;
                if (best.key != kNoTriangle) pos = posOf[0xfffffffeu - best.key];   // per-lane gather
                // A kept weight that is exactly zero (the hit lies on an edge of the triangle) may carry the other sign in a
                // class form (pttri.h): those lanes — hardly ever one — take the general form's weights, so that even the sign
                // of a zero normal component is the reference's. The general form accepts the same hit at the same distance.
                const bool zeroWeight = best.key != kNoTriangle && (best.w1 == 0.0f || best.w2 == 0.0f);
                if (waveAny(zeroWeight)) {
                    if (zeroWeight) {
                        const float4* rows = sc + L.offTri + 3 * pos;   // per-lane gathers
                        const float4* prim = sc + L.offPrimTri + 2 * pos;
                        const pttri::Head g = pttri::head<0, 0, kPrimary>(xyz(rows[0]), xyz(rows[1]), xyz(rows[2]), xyz(prim[0]), xyz(prim[1]), prim[0].w, o, d);
                        float b0;
                        pttri::weights<0, 0>(g, d, b0, best.w1, best.w2);
                    }
                }
                if (best.key != kNoTriangle) {
                    h.distance = best.dist;
                    h.kind = 2;
                    h.idx = pos;
                    h.w1 = best.w1;
                    h.w2 = best.w2;
                    h.w0 = 1.0f - (best.w1 + best.w2);  // Primitives.h:64, from the kept pair
                }
            }
            return h;
        }
    } else if (L.triDetBounded && waveAll(dot(d, d) < 0x1p30f)) {
        // the caller's order, the general body, the sequential rule; the reciprocal's range guard proven once per query
        TriBest best{h.distance, kNoTriangle, 0.0f, 0.0f};
        for (int i = 0; i < L.numTriangles; ++i)
            triangleClassed<kPrimary, 0, 0, false>(sc + L.offTri + 3 * i, sc + L.offPrimTri + 2 * i, (uint32_t)i, o, d, liveMask, best);
        if (best.key != kNoTriangle) {
            h.distance = best.dist;
            h.kind = 2;
            h.idx = (int)best.key;
            h.w1 = best.w1;
            h.w2 = best.w2;
            h.w0 = 1.0f - (best.w1 + best.w2);  // Primitives.h:64, from the kept pair
        }
        return h;
    }
    const int* posOfOriginal = reinterpret_cast<const int*>(sc + L.offTriPos);
    for (int k = 0; k < L.numTriangles; ++k) {   // the guarded loop, in the caller's order: unbounded edges, or a ray of enormous length
        const int i = L.triClassed ? posOfOriginal[k] : k;   // where original triangle k is stored
        const TriRows tcur = kPrimary ? loadTriEdges(sc + L.offTri + 3 * i) : loadTri(sc + L.offTri + 3 * i);
        const TriHit th = kPrimary ? triangleTestPrimary(tcur, sc[L.offPrimTri + 2 * i], loadRow16(sc + L.offPrimTri + 2 * i + 1), d,
                                                         h.distance, liveMask)
                                   : triangleTest(tcur, o, d, h.distance, liveMask);
        if (th.hit) {
            h.distance = th.dist;
            h.kind = 2;
            h.idx = i;
            h.w0 = th.w0;
            h.w1 = th.w1;
            h.w2 = th.w2;
        }
    }
    return h;
}

// the triangle half of lineOfSight for a wave whose lanes all test the same triangle at a time: `need` = lanes that still want an
// answer, `blocked` collects the verdicts. Grouped storage (SceneLayout::triClassed): one loop per edge class with its shorter
// body, the reciprocal's guard proven once per pass; otherwise, and for non-finite or enormous segments, the guarded general test.
__device__ __forceinline__ void anyTriangleLoop(const float4* sc, const SceneLayout& L, vec3 lo, vec3 w_i, float distance, unsigned long long& need,
                                                unsigned long long& blocked) {
    if (L.triClassed && classedQueryOk(lo, w_i)) {
#define PTSS_ANY_BODY(c1, c2, t)   \
    if (need == 0ull) break;      \
    triangleClassedAny<c1, c2>(sc + L.offTri + 3 * t, lo, w_i, distance, need, blocked);
        PTSS_FOR_TRIANGLES_BY_CLASS(L, PTSS_ANY_BODY);
#undef PTSS_ANY_BODY
        return;
    }
    for (int i = 0; i < L.numTriangles; ++i) {
        if (need == 0ull) break;
        const TriRows tcur = loadTri(sc + L.offTri + 3 * i);
        const TriHit th = triangleTest(tcur, lo, w_i, distance, need);
        blocked |= th.hitMask;
        need &= ~th.hitMask;
    }
}

// the triangle half of lineOfSight alone (the sphere half having been answered by anySpheresHybrid)
__device__ __forceinline__ bool anyTriangles(const float4* sc, const SceneLayout& L, vec3 lo, vec3 w_i, float distance, bool live) {
    unsigned long long need = maskOf(live), blocked = 0ull;
    anyTriangleLoop(sc, L, lo, w_i, distance, need, blocked);
    return __builtin_amdgcn_inverse_ballot_w64(blocked);
}

// ---- the any-hit loops of lineOfSight, CudaTracer.cu:437-452: true when some primitive blocks the
// segment. Order-independent (the reference returns at the first accepted primitive and no test
// depends on another). `live`: this lane carries a segment. -----------------------------------------
template <bool kAccel, bool kBounded>
__device__ __forceinline__ bool anyHit(const float4* sc, const SceneLayout& L, vec3 lo, vec3 w_i, float distance,
                                       bool live) {
    bool occluded = false;
    if constexpr (kAccel) occluded = anySphereChunked(sc, L, lo, w_i, distance, live);
    for (int base = 0; base < (kAccel ? 0 : L.numSpheres); base += 32) {
        const int cnt = (L.numSpheres - base < 32) ? (L.numSpheres - base) : 32;
        uint32_t mask = sphereCandidatesPairs<kBounded>(sc + L.offSphere + base, cnt, lo, w_i);
        mask &= (live && !occluded) ? lowBits(cnt) : 0u;
        PTSS_DIAG_CANDIDATES(mask, live, 4);
        while (mask != 0) {
            const int j = __builtin_ctz(mask);
            mask &= mask - 1;
            float t;
            if (sphereTest(sc[L.offSphere + base + j], lo, w_i, distance, t)) {
                occluded = true;
                mask = 0;
            }
        }
    }
    unsigned long long need = maskOf(live) & ~maskOf(occluded);  // lanes that still want an answer
    unsigned long long blocked = 0ull;
    anyTriangleLoop(sc, L, lo, w_i, distance, need, blocked);
    return occluded || __builtin_amdgcn_inverse_ballot_w64(blocked);
}

// ---- the same any-hit with the primitive list SPLIT over g = 1 << shift lanes per segment: lane `sub` of a
// segment's group visits primitives sub, sub + g, sub + 2g, ...; the caller ORs the group's verdicts. Every test is
// the scalar test on the same operands, and lineOfSight's answer is an OR over independent tests, so the verdict is
// the one anyHit gives. Used when a pass over the wave's queue holds fewer than 64 segments: 8 segments x 8 lanes
// cost an eighth of a dense pass instead of a whole one. Rows are gathered per lane here (no broadcast). ------------
template <bool kBounded>
__device__ __forceinline__ bool anyHitSplit(const float4* sc, const SceneLayout& L, vec3 lo, vec3 w_i, float distance,
                                            bool live, int shift, int sub) {
    bool occluded = false;
    const int g = 1 << shift;
    const int sphereSteps = (L.numSpheres + g - 1) >> shift;
    for (int base = 0; base < sphereSteps; base += 32) {
        const int cnt = (sphereSteps - base < 32) ? (sphereSteps - base) : 32;
        uint32_t mask = sphereCandidatesStridedPairs<kBounded>(sc + L.offSphere + (base << shift) + sub, g, cnt, lo, w_i);
        // this lane's spheres are sub, sub + g, ...: step j exists for it iff (j << shift) + sub < numSpheres
        mask &= (live && !occluded) ? lowBitsClamped(((L.numSpheres - sub + g - 1) >> shift) - base) : 0u;
        while (mask != 0) {
            const int j = __builtin_ctz(mask);
            mask &= mask - 1;
            float t;
            if (sphereTest(sc[L.offSphere + ((base + j) << shift) + sub], lo, w_i, distance, t)) {
                occluded = true;
                mask = 0;
            }
        }
    }
    const int triSteps = (L.numTriangles + g - 1) >> shift;
    unsigned long long need = maskOf(live) & ~maskOf(occluded);
    unsigned long long blocked = 0ull;
    for (int k = 0; k < triSteps; ++k) {
        if (need == 0ull) break;
        const int idx = (k << shift) + sub;
        const bool in = idx < L.numTriangles;
        const TriRows tcur = loadTri(sc + L.offTri + 3 * (in ? idx : 0));
        const TriHit th = triangleTest(tcur, lo, w_i, distance, need & maskOf(in));
        blocked |= th.hitMask;
        need &= ~th.hitMask;
    }
    return occluded || __builtin_amdgcn_inverse_ballot_w64(blocked);
}

// ---- lineOfSight for the TWO segments a surface point sends to the two lights of an NEE round. They share their origin,
// and so everything the tests compute from origin and primitive alone: a sphere's v = o - centre and c = |v|^2 - r^2
// (7 of its 13 / 15 instructions), a triangle's s = o - v0, r = s x e1 and e2 . r (12 of the ~32 up to the distance test).
// Each segment's own part is the scalar test's, on the same operands in the same order, so the two verdicts are the ones
// two separate queue entries would get. kSplit: 1 << shift lanes share an entry, lane `sub` takes primitives sub, sub + g, ...
// (anyHitSplit's scheme); otherwise one lane per entry and broadcast rows. liveA / liveB: the segment exists and is needed.
template <bool kBounded, bool kSplit>
__device__ __forceinline__ void pairAnyHit(const float4* sc, const SceneLayout& L, vec3 lo, vec3 wA, float dA, bool liveA, vec3 wB, float dB,
                                           bool liveB, int shift, int sub, bool& occA, bool& occB) {
    occA = false;
    occB = false;
    const int g = kSplit ? (1 << shift) : 1;
    const int sphereSteps = kSplit ? ((L.numSpheres + g - 1) >> shift) : L.numSpheres;
    for (int base = 0; base < sphereSteps; base += 32) {
        const int cnt = (sphereSteps - base < 32) ? (sphereSteps - base) : 32;
        const int trips = (cnt + 1) >> 1;
        uint32_t revA = 0, revB = 0;
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int step = base + 2 * t + u;
                const float4 sp = kSplit ? sc[L.offSphere + ((step << shift) + sub)] : sc[L.offSphere + step];
                const vec3 v = lo - xyz(sp);
                const float c = dot(v, v) - sp.w;
                const float hA = dot(wA, v), hB = dot(wB, v);
                if constexpr (kBounded) {
                    shiftInMayHit(revA, hA * hA, c);
                    shiftInMayHit(revB, hB * hB, c);
                } else {
                    const float c4 = 4 * c, bA = hA * 2, bB = hB * 2;
                    shiftInMayHit(revA, bA * bA, c4);
                    shiftInMayHit(revB, bB * bB, c4);
                }
            }
        }
        const uint32_t valid = kSplit ? lowBitsClamped(((L.numSpheres - sub + g - 1) >> shift) - base) : lowBits(cnt);
        uint32_t maskA = (__builtin_bitreverse32(revA) >> (32 - 2 * trips)) & ((liveA && !occA) ? valid : 0u);
        uint32_t maskB = (__builtin_bitreverse32(revB) >> (32 - 2 * trips)) & ((liveB && !occB) ? valid : 0u);
        while (maskA != 0) {
            const int j = __builtin_ctz(maskA);
            maskA &= maskA - 1;
            float t;
            if (sphereTest(sc[L.offSphere + (kSplit ? (((base + j) << shift) + sub) : (base + j))], lo, wA, dA, t)) {
                occA = true;
                maskA = 0;
            }
        }
        while (maskB != 0) {
            const int j = __builtin_ctz(maskB);
            maskB &= maskB - 1;
            float t;
            if (sphereTest(sc[L.offSphere + (kSplit ? (((base + j) << shift) + sub) : (base + j))], lo, wB, dB, t)) {
                occB = true;
                maskB = 0;
            }
        }
    }
    unsigned long long needA = __ballot(liveA && !occA), needB = __ballot(liveB && !occB);
    unsigned long long blockedA = 0ull, blockedB = 0ull;
    const int triSteps = kSplit ? ((L.numTriangles + g - 1) >> shift) : L.numTriangles;
    if constexpr (!kSplit) {   // every lane at the same triangle: one loop per edge class (grouped storage), origin part shared
        if (L.triClassed && classedQueryOk(lo, wA) && waveAll(dot(wB, wB) < 0x1p30f)) {
#define PTSS_PAIR_BODY(c1, c2, t)            \
    if ((needA | needB) == 0ull) break;     \
    triangleClassedPair<c1, c2>(sc + L.offTri + 3 * t, lo, wA, dA, wB, dB, needA, needB, blockedA, blockedB);
            PTSS_FOR_TRIANGLES_BY_CLASS(L, PTSS_PAIR_BODY);
#undef PTSS_PAIR_BODY
            occA = occA || __builtin_amdgcn_inverse_ballot_w64(blockedA);
            occB = occB || __builtin_amdgcn_inverse_ballot_w64(blockedB);
            return;
        }
    }
    for (int k = 0; k < triSteps; ++k) {
        if ((needA | needB) == 0ull) break;
        const int idx = kSplit ? ((k << shift) + sub) : k;
        const bool in = !kSplit || idx < L.numTriangles;
        const TriRows tr = loadTri(sc + L.offTri + 3 * (in ? idx : 0));
        const unsigned long long inMask = kSplit ? maskOf(in) : ~0ull;
        const vec3 v0 = xyz(tr.a), e1 = xyz(tr.b), e2 = xyz(tr.c);
        const vec3 sv = lo - v0;               // shared by the two segments (Primitives.h:46-49)
        const vec3 r = cross(sv, e1);
        const float e2r = dot(e2, r);
        if (needA != 0ull) {
            const vec3 q = cross(wA, e2);
            const float det = dot(e1, q);
            const float inverseDet = triRcp(det);
            const float dist = e2r * inverseDet;
            const unsigned long long pass = needA & inMask & maskOf(!(ptm::abs(det) <= 1e-7f)) & maskOf(!(dist <= 0.0f)) & maskOf(!(dist > dA));
            if (pass != 0ull) {
                const float b1 = dot(sv, q) * inverseDet;
                const float b2 = dot(wA, r) * inverseDet;
                const float b0 = 1.0f - (b1 + b2);
                const unsigned long long hit = pass & maskOf(!(b0 < 0)) & maskOf(!(b1 < 0)) & maskOf(!(b2 < 0));
                blockedA |= hit;
                needA &= ~hit;
            }
        }
        if (needB != 0ull) {
            const vec3 q = cross(wB, e2);
            const float det = dot(e1, q);
            const float inverseDet = triRcp(det);
            const float dist = e2r * inverseDet;
            const unsigned long long pass = needB & inMask & maskOf(!(ptm::abs(det) <= 1e-7f)) & maskOf(!(dist <= 0.0f)) & maskOf(!(dist > dB));
            if (pass != 0ull) {
                const float b1 = dot(sv, q) * inverseDet;
                const float b2 = dot(wB, r) * inverseDet;
                const float b0 = 1.0f - (b1 + b2);
                const unsigned long long hit = pass & maskOf(!(b0 < 0)) & maskOf(!(b1 < 0)) & maskOf(!(b2 < 0));
                blockedB |= hit;
                needB &= ~hit;
            }
        }
    }
    occA = occA || __builtin_amdgcn_inverse_ballot_w64(blockedA);
    occB = occB || __builtin_amdgcn_inverse_ballot_w64(blockedB);
}

// one light's Lambert term, CudaTracer.cu:360-366 / :379-385
__device__ __forceinline__ void addLambertTerm(vec3& radiance, float cosI, vec3 power, float distance2,
                                               float4 diffuse /* colour, diffAvg */) {
    const vec3 L_i = power / (float)(4 * ptm::kPi * distance2);
    radiance.x += cosI * L_i.x * diffuse.x * diffuse.w * ptm::kInvPi;
    radiance.y += cosI * L_i.y * diffuse.y * diffuse.w * ptm::kInvPi;
    radiance.z += cosI * L_i.z * diffuse.z * diffuse.w * ptm::kInvPi;
}

// CudaTracer.cu:579-585
__device__ __forceinline__ quat rotateVectorToVector(vec3 source, vec3 target) {
    const vec3 axis = cross(source, target);
    return normalize(q4(1.0f + dot(source, target), axis.x, axis.y, axis.z));
}

// ---- computeIndirectRadianceAndScatter, CudaTracer.cu:208-318 ---------------------------------
// The three random-direction samplers of the reference (Lambert :533-545, Phong :547-559, Beckmann :561-577) all
// draw two uniforms and end the same way: a vector (a*cos(az), y, a*sin(az)) about +Y, rotated onto the lobe axis by
// rotateVectorToVector (:579-585). With 64 incoherent rays nearly every wave holds lanes of all three kinds, so the
// lobe CHOICE runs divergently (it is cheap) and the draws + sincos + rotation run ONCE, for all sampling lanes
// together; each lane performs exactly the operations, in the order, that its own sampler performs in the
// reference (the draws keep their order: Lambert/Phong use the first for the azimuth and the second for the
// elevation, Beckmann the first for the elevation and the second for the azimuth).
enum LobeKind { kLobeNone = 0, kLobeLambert = 1, kLobePhong = 2, kLobeBeckmann = 3 };

__device__ __forceinline__ vec3 scatter(const float4* mat, RayRegs& ray, vec3 point, vec3 normal, float cosI) {
    const float4 mDiffuse = mat[0];   // diffuseColor, diffAvg
    const float4 mSpecular = mat[1];  // specularColor, specAvg
    const float4 mMisc = mat[4];      // specularExponent, indexOfRefraction, flags
    const float refrAvg = mat[2].w;
    const int flags = (int)asU(mMisc.z);

    float r = ptrng::uniform(ray.rng);
    PTSS_DIAG_SCATTER(0, true);  // waves (and lanes) in scatter at all

    int kind = kLobeNone;
    bool decided = false;
    vec3 axis = normal;
    vec3 result = v3(0, 0, 0);
    const vec3 incident = ray.d;

    if (mDiffuse.w > 0.0f) {
        r -= mDiffuse.w;
        if (r < 0.0f) {  // randomDirectionLambert about the normal
            ray.o = point + ptm::kRayBump * normal;
            kind = kLobeLambert;
            decided = true;
            result = xyz(mDiffuse);
        }
    }

    PTSS_DIAG_SCATTER(1, !decided);  // the non-Lambert block
    if (!decided) {
        // computeSinT2AndRefractiveIndexes :474-494 (flips cosI when inside)
        float n1, n2;
        if (cosI > 0) {
            n2 = mMisc.y;
            n1 = 1.0f;
        } else {
            cosI = -cosI;
            n1 = mMisc.y;
            n2 = 1.0f;
        }
        // The Snell / Fresnel terms (two square roots' worth and three divisions) feed only the Fresnel-weighted specular
        // lobe (:248-249) and the refraction lobe (:300-311). A material with the pure-reflection bit (mirrors AND every
        // Cook-Torrance material, 0x03 & 0x01) and no refraction never reads them: its lanes skip the block, and a wave
        // without glass skips it altogether. (cosI's flip above is kept: reflRay uses it.)
        const bool readsFresnel = (mSpecular.w > 0.0f && !(flags & PTSS_MAT_FLAG_PURE_REFLECTION)) || refrAvg > 0.0f;
        float n = 0.0f, sinT2 = 0.0f;
        float fresnelReflective = 1.0f;
        PTSS_DIAG_SCATTER(2, readsFresnel);  // Snell / Fresnel terms
        if (readsFresnel) {
            n = ptm::div(n1, n2);    // computeSinT2AndRefractiveIndexes :491-493
            sinT2 = n * n * (1.0f - cosI * cosI);
        }
        if (readsFresnel && !(sinT2 > 1.0f)) {   // computeFresnelForReflectance :457-472
            const float cosT = ptm::sqrt(1.0f - sinT2);
            const float r_s = ptm::div(n1 * cosI - n2 * cosT, n1 * cosI + n2 * cosT);
            const float r_p = ptm::div(n2 * cosI - n1 * cosT, n2 * cosI + n1 * cosT);
            fresnelReflective = (r_s * r_s + r_p * r_p) * 0.5f;
        }

        if (mSpecular.w > 0.0f) {
            if (flags & PTSS_MAT_FLAG_PURE_REFLECTION)
                r -= mSpecular.w;
            else
                r -= mSpecular.w * fresnelReflective;

            if (r < 0.0f) {
                decided = true;
                if (flags & PTSS_MAT_FLAG_COOK_TORRANCE) {
                    kind = kLobeBeckmann;  // micro-normal about the surface normal; the reflection follows below
                } else {
                    // reflRay(ray, surfel, cosI) :496-503
                    ray.d = ray.d - (2 * (-cosI)) * normal;
                    ray.o = point + (normal * ptm::kRayBump);
                    if (mMisc.x != ptm::inf()) {  // randomDirectionPhong about the mirror direction
                        kind = kLobePhong;
                        axis = ray.d;
                    }
                    result = xyz(mSpecular);
                }
            }
        }

        if (!decided && refrAvg > 0.0f) {
            const float fresnelRefractive = 1.0f - fresnelReflective;
            r -= refrAvg * fresnelRefractive;
            PTSS_DIAG_SCATTER(3, r < 0.0f);  // refraction lobe
            if (r < 0.0f) {
                // refrRay :516-531
                decided = true;
                if (sinT2 > 1.0f) ray.active = false;
                const float cosT = ptm::sqrt(1.0f - sinT2);
                const vec3 w_o = normalize(n * ray.d + (n * cosI - cosT) * normal);
                ray.o = point + (w_o * ptm::kRayBump);
                ray.d = w_o;
                result = v3(1, 1, 1);
            }
        }

        if (!decided) ray.active = false;  // absorbed, :316-317
    }

    PTSS_DIAG_SCATTER(4, kind != kLobeNone);       // the shared sampler tail
    PTSS_DIAG_SCATTER(5, kind == kLobeBeckmann);   // ... with the Beckmann elevation (atan, log) and the Cook-Torrance weight
    PTSS_DIAG_SCATTER(6, kind == kLobePhong);      // ... with the Phong elevation (pow)
    PTSS_DIAG_SCATTER(7, kind == kLobeLambert);
    if (kind != kLobeNone) {  // one copy of the sampler for every kind
        const float u1 = ptrng::uniform(ray.rng);
        const float u2 = ptrng::uniform(ray.rng);
        float azimuth, a, y;
        if (kind == kLobeBeckmann) {
            const float roughness = mat[3].w;
            const float theta = ptm::atan(-roughness * roughness * ptm::log(1.0f - u1));  // :564
            azimuth = u2 * 2 * ptm::kPi;                                                      // :565
            ptm::sincos(theta, a, y);  // m = (sinTheta * cosPhi, cosTheta, sinTheta * sinPhi), :567-569
        } else {
            azimuth = u1 * 2 * ptm::kPi;                                                      // :536, :550
            y = (kind == kLobeLambert) ? ptm::sqrt(u2) : ptm::pow(u2, ptm::rcp(mMisc.x + 1));  // :537-538, :551-552
            a = ptm::sqrt(1 - y * y);                                                          // :539, :553
        }
        float sn, cs;
        ptm::sincos(azimuth, sn, cs);
        const vec3 sampled = rotate(rotateVectorToVector(v3(0, 1, 0), axis), v3(a * cs, y, a * sn));
        if (kind == kLobeBeckmann) {
            const vec3 beckmannNormal = sampled;
            // reflRay(ray, point, normal) :505-514
            const float cosB = ptm::abs(dot(ray.d, beckmannNormal));
            ray.d = ray.d - (2 * (-cosB)) * beckmannNormal;
            ray.o = point + (beckmannNormal * ptm::kRayBump);

            const vec3 half = normalize(ray.d - incident);
            const float nh = ptm::abs(dot(normal, half));
            const float nl = ptm::abs(dot(normal, ray.d));
            const float vh = ptm::abs(dot(incident, half));
            const float nv = ptm::abs(cosI);
            const float geometric = ptm::min(ptm::min(1.0f, ptm::div(2 * nh * nl, vh)), ptm::div(2 * nh * nv, vh));
            result = xyz(mSpecular) * geometric / nv;
        } else {
            ray.d = sampled;
        }
    }
    return result;
}

// one channel of writeToPixelsKernel, CudaTracer.cu:72-85: clamp, gamma 1/2.2, scale to 8 bits — in the proven-equal table
// form (ptquant.h): a hardware log2/exp2 guess settled by two exact threshold compares, ~12 instructions instead of the ~90
// of the software pow; three of these run for every wave that ends a path.
__device__ __forceinline__ uint32_t quantizeSample(float radiance, const float* T) { return ptq::quantize_fast(radiance, T); }

// The per-pixel home record of the random stream: 8 words (v0..v4, d, 2 pad) = one 32-byte sector, so
// parking or fetching a stream is two 16-byte accesses instead of six scattered 4-byte ones.
__device__ __forceinline__ void loadHome(const uint32_t* __restrict__ home, uint32_t p, ptrng::State& s) {
    const uint4 a = reinterpret_cast<const uint4*>(home)[2 * p];
    const uint4 b = reinterpret_cast<const uint4*>(home)[2 * p + 1];
    s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w;
    s.v[4] = b.x; s.d = b.y;
}
__device__ __forceinline__ void storeHome(uint32_t* __restrict__ home, uint32_t p, const ptrng::State& s) {
    reinterpret_cast<uint4*>(home)[2 * p] = uint4{s.v[0], s.v[1], s.v[2], s.v[3]};
    reinterpret_cast<uint4*>(home)[2 * p + 1] = uint4{s.v[4], s.d, 0u, 0u};
}

struct U3 {  // one totalPixelColors entry, moved as a single 12-byte access
    uint32_t x, y, z;
};

// A path ended: writeToPixelsKernel for this ray (CudaTracer.cu:63-104) + park the RNG stream.
// S == 1: the reference's read-modify-write of totalPixelColors and the display pixel, right here (one writer per pixel).
// S > 1: several lanes of a launch may end paths of the SAME pixel, so the tone-mapped 8-bit sample is parked in the
// stream's own word instead and displayKernel adds the S words of each pixel into the accumulator when the pass is
// complete (integer sums: order-free, still exact); the float sum is kept per stream (summed in lane order on read).
__device__ __forceinline__ void finishPath(const FrameBuffers& fb, const RayRegs& r, const float* quantT) {
    const uint32_t p = pixOf(r.pix), lane = laneOf(r.pix);
    const uint32_t stream = lane * fb.plane + p;
    const uint32_t qx = quantizeSample(r.L0.x, quantT), qy = quantizeSample(r.L0.y, quantT), qz = quantizeSample(r.L0.z, quantT);
    if (fb.samples == 1) {
        U3* acc = reinterpret_cast<U3*>(fb.accum) + p;
        U3 t = *acc;
        t.x += qx;
        t.y += qy;
        t.z += qz;
        *acc = t;
        if (fb.pixels) {
            const uint32_t px = (uint32_t)(unsigned char)(t.x * fb.inverseTicks + 0.5f) |
                                ((uint32_t)(unsigned char)(t.y * fb.inverseTicks + 0.5f) << 8) |
                                ((uint32_t)(unsigned char)(t.z * fb.inverseTicks + 0.5f) << 16) | (255u << 24);
            reinterpret_cast<uint32_t*>(fb.pixels)[p] = px;  // uchar4 {x, y, z, w = 255}
        }
    } else {
        // S > 1: every stream ends exactly one path per pass, so its 8-bit sample goes to the stream's own word with a
        // plain store; displayKernel adds the S words of a pixel into the accumulator at the end of the pass. (Three
        // atomics per path instead cost 34 % of the last-bounce kernel, where every ray finishes at once.)
        fb.staged[stream] = qx | (qy << 8) | (qz << 16);
    }
    if (fb.fsum) {
        float* fs = fb.fsum + 3u * stream;
        fs[0] += r.L0.x;
        fs[1] += r.L0.y;
        fs[2] += r.L0.z;
    }
    storeHome(fb.rngHome, stream, r.rng);
}

// ---- the loop guard with frame lanes (FrameBuffers, "frame lanes"): the frame's live count of bounce b >= 1 when this
// lane's own count `own` is not above the threshold. Waits (bounded) until every workgroup of each peer's bounce b - 1 has
// ended (the peer's done counters reach `target[p]`), then adds the peer's sixteen shard counters of bounce b.
// Called by at most one workgroup per shard of a lane that holds <= 128 rays, and by flushKernel.
// Every wait for a peer lane is bounded by TIME — about two seconds of the 100 MHz real-time counter (s_memrealtime), whatever
// a poll costs under load —: a peer stream that never runs must not hang the device. A wait that expires counts itself in
// guardTimeouts, which the host turns into PTSS_ETIMEOUT at its next synchronising call (ptss_api.hip checkLaneTimeouts).
constexpr unsigned long long kPeerWaitTicks = 200000000ull;
__device__ __forceinline__ bool peerWaitExpired(unsigned long long& since) {
    const unsigned long long now = wall_clock64();
    if (since == 0ull) {   // the first unsuccessful poll starts the clock
        since = now | 1ull;
        return false;
    }
    return now > since && now - since > kPeerWaitTicks;
}

__device__ __forceinline__ uint32_t frameLiveCount(const FrameBuffers& fb, int bounce, uint32_t own, const uint32_t* target) {
    uint32_t total = own;
    for (uint32_t p = 0; p < fb.numPeers; ++p) {
        unsigned long long since = 0ull;
        for (;;) {
            uint32_t ended = 0;
            for (int s = 0; s < kShards; ++s)
                ended += __hip_atomic_load(fb.peerDone[p] + countIndex(bounce - 1, s), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            // (>=, wrap-safe: lanes may run up to one frame apart, and a peer that is ahead has added its next frame's
            // workgroups already; its counts of THIS frame stay intact meanwhile — they live in the buffer its flushKernel
            // re-arms only after this lane's frame)
            if ((int32_t)(ended - target[p]) >= 0) break;
            __builtin_amdgcn_s_sleep(64);
            if (peerWaitExpired(since)) {
                if (threadIdx.x == 0) atomicAdd(fb.guardTimeouts, 1u);
                break;
            }
        }
        for (int s = 0; s < kShards; ++s)
            total += __hip_atomic_load(fb.peerCounts[p] + countIndex(bounce, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return total;
}

// ---- LDS work area behind the scene image -----------------------------------------------------
// block: [0..kWaves) wave survivor totals, [8] block base in the output region
// per wave: the shadow-ray queue of one NEE round (kNeeLights lights x 64 lanes):
//           7 float planes (lo.xyz, w_i.xyz, max distance) + 1 word (owner lane | slot-in-round << 8),
//           then kNeeLights x 64 answer BYTES.
constexpr int kNeeLights = 2;                       // lights regrouped per round
constexpr int kQueueCap = kNeeLights * 64;
static_assert(kQueueCap == kQueueCapConst, "anySpheresHybrid's plane stride");
constexpr int kWaveLdsWords = 8 * kQueueCap + kNeeLights * 64 / 4;  // answers are bytes: 24,048 -> 22,512 B per workgroup
                                                                    // with the 38-primitive scenes, i.e. 7 workgroups per CU instead of 6
constexpr int kBlockScratchVec4 = 4;
constexpr int kBlockLdsVec4 = kBlockScratchVec4 + (kWaves * kWaveLdsWords + 3) / 4;

}  // namespace

// =================================================================================================
// curand_init(seed, sequence, 0, ...) per stream: sequence = globalPixel * S + lane (S = 1: the pixel index,
// as the reference's `offset`, CudaTracer.cu:26-28)
__global__ void rngInitKernel(uint32_t* __restrict__ rngHome, uint32_t plane, uint32_t samples, TileMap tile, uint64_t seed,
                              const uint32_t* __restrict__ jumpTable) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)tile.width * (uint32_t)tile.localRows;
    const uint32_t lane = i / plane, p = i - lane * plane;
    if (lane >= samples || p >= n) return;
    const PixelCoord pc = locate(tile, p);
    ptrng::State s = ptrng::seeded(seed);
    ptrng::skip_subsequences(s, pc.globalIndex * samples + lane, jumpTable);
    storeHome(rngHome, i, s);
}

__global__ void clearKernel(FrameBuffers fb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fb.numPixels) return;
    fb.accum[3 * i] = 0;
    fb.accum[3 * i + 1] = 0;
    fb.accum[3 * i + 2] = 0;
    if (fb.fsum)
        for (uint32_t l = 0; l < fb.samples; ++l) {
            float* fs = fb.fsum + 3u * (l * fb.plane + i);
            fs[0] = 0;
            fs[1] = 0;
            fs[2] = 0;
        }
    if (fb.pixels) fb.pixels[i] = ptss_uchar4{0, 0, 0, 0};
}

// S > 1 only: the display value of writeToPixelsKernel (CudaTracer.cu:94-98), once the pass has added all its samples
__global__ void displayKernel(FrameBuffers fb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fb.numPixels) return;
    U3 t = reinterpret_cast<const U3*>(fb.accum)[p];
    uint32_t l = 0;
    for (; l + 8 <= fb.samples; l += 8) {  // this pass's S samples of the pixel (finishPath), coalesced per lane plane; eight
        uint32_t q[8];                      // independent fetches in flight per thread (one at a time ran at 2.6 TB/s)
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = fb.staged[(l + k) * fb.plane + p];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t.x += q[k] & 255u;
            t.y += (q[k] >> 8) & 255u;
            t.z += (q[k] >> 16) & 255u;
        }
    }
    for (; l < fb.samples; ++l) {
        const uint32_t q = fb.staged[l * fb.plane + p];
        t.x += q & 255u;
        t.y += (q >> 8) & 255u;
        t.z += (q >> 16) & 255u;
    }
    reinterpret_cast<U3*>(fb.accum)[p] = t;
    if (fb.pixels) {
        const uint32_t px = (uint32_t)(unsigned char)(t.x * fb.inverseTicks + 0.5f) |
                            ((uint32_t)(unsigned char)(t.y * fb.inverseTicks + 0.5f) << 8) |
                            ((uint32_t)(unsigned char)(t.z * fb.inverseTicks + 0.5f) << 16) | (255u << 24);
        reinterpret_cast<uint32_t*>(fb.pixels)[p] = px;
    }
}

// Origin-only parts of the primary-ray tests, one thread per primitive; rerun when the camera moves.
__global__ void primaryPrepKernel(float4* __restrict__ blob, SceneLayout L, vec3 origin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L.numSpheres && !L.accelSpheres) {
        const float4 sp = blob[L.offSphere + i];
        const vec3 v = origin - xyz(sp);
        blob[L.offPrimSphere + i] = float4{v.x, v.y, v.z, dot(v, v) - sp.w};
    }
    if (L.accelSpheres && i < L.numChunks) {   // the chunk test's origin part (shiftInChunkPrimary)
        const float4 b = blob[L.offChunk + i];
        const vec3 v = origin - xyz(b);
        const float x = dot(v, v) - b.w;
        blob[L.offPrimChunk + i] = float4{v.x, v.y, v.z, x - ptm::abs(x) * 2.4e-7f};   // down by more than an ulp
    }
    if (i < L.numTriangles) {
        const vec3 v0 = xyz(blob[L.offTri + 3 * i]), e1 = xyz(blob[L.offTri + 3 * i + 1]), e2 = xyz(blob[L.offTri + 3 * i + 2]);
        const vec3 s = origin - v0;
        const vec3 r = cross(s, e1);
        blob[L.offPrimTri + 2 * i] = float4{s.x, s.y, s.z, dot(e2, r)};
        blob[L.offPrimTri + 2 * i + 1] = float4{r.x, r.y, r.z, 0.0f};
    }
}

// kSceneInLds = true : the scene blob is staged into LDS once per workgroup and read by broadcast
//                      (ds_read, same address in every lane; per-lane gathers in the candidate loops).
// kSceneInLds = false: the blob is read in place (scalar loads where the address is wave-uniform);
//                      scenes whose image does not fit the 64 KiB dynamic-LDS window take this path.
//
// One tile = kBlock rays = one workgroup pass:
//   1. load ray, closest hit (sphere candidate masks + uniform triangle loop), surfel, emission;
//   2. next-event estimation, kNeeLights lights per round: every lane that hit a front face draws
//      the light samples (RNG order as the reference); lanes whose Lambert term can be non-zero
//      append a shadow segment to their wave's LDS queue; the wave then traces the queue densely
//      (lane k takes entry k, k+64, ...) and posts the answers back through LDS. No barrier:
//      producer and consumer are the same wave.
//   3. scatter (lobe choice + new direction), Beer-Lambert, radiance update;
//   4. paths that ended tone-map into the accumulator; 5. survivors are compacted into the
//      shard's region of the other pool.
//
// Exactness of the shadow-ray skip (step 2): the reference adds cosI*L_i*diffuseColor*diffAvg/pi
// when the light is visible. With diffAvg == 0 or cosI == 0 that term is +-0 whenever L_i is finite
// (distance2 in (0, inf); light powers and diffuse colours are checked finite at ptss_create,
// SceneLayout::neeSkipSafe), and radiance + (+-0) == radiance, so visibility cannot change the
// result and the segment is not traced. Every other case runs the literal path.
//
// kFirst: bounce 0 makes its own rays — computeEyeRaysKernel (CudaTracer.cu:51-61, 321-343) is fused in: the
// lane fetches its pixel's random stream from the home record, draws the two jitter samples, builds the
// eye ray in registers (no ray pool read) and intersects with the camera-origin precomputes.
// kAccel: the scene image carries the chunked sphere structure (SceneLayout::accelSpheres) — its own instantiations, so
// that scenes without it run exactly the code they ran before.
// What one workgroup needs to trace one tile (bounceTile): the staged scene, this bounce's input and output regions of its
// shard, the wave's LDS queue. Built by bounceBody (one launch per bounce) and by frameKernel (one launch per frame).
struct TileEnv {
    const float4* sc;          // the scene image as the tile reads it (LDS or global)
    const float4* sceneBlob;   // ... in global memory (the many-sphere image's cold integer tables)
    const float* quantT;
    const float* in;           // this shard's region of the input pool
    float* out;                // ... of the output pool
    float* wq;                 // this wave's LDS queue
    uint32_t* wqOwner;
    unsigned char* wqAnswer;
    uint32_t shard, lane, n;   // n: rays of this shard entering the bounce
    int numLights, bounce;
};

// One tile = kBlock rays of bounce env.bounce, starting at slot / frame-tile offset `base` of the shard. kCoherentIo: the ray
// pools are read and written past the L1 (ldPlane<true>): other workgroups of the SAME launch produced / will consume them.
template <bool kLast, bool kFirst, bool kAccel, bool kBounded, bool kPairs, bool kCoherentIo>
__device__ __forceinline__ void bounceTile(const FrameBuffers& fb, const SceneLayout& L, const TileMap& tile, const EyeParams& eye, const TileEnv& env,
                                           uint32_t base) {
    const float4* sc = env.sc;
    const float4* sceneBlob = env.sceneBlob;
    const float* quantT = env.quantT;
    const float* __restrict__ in = env.in;
    float* __restrict__ out = env.out;
    float* wq = env.wq;
    uint32_t* wqOwner = env.wqOwner;
    unsigned char* wqAnswer = env.wqAnswer;
    const uint32_t shard = env.shard, lane = env.lane, n = env.n;
    const int numLights = env.numLights, bounce = env.bounce;
    {
        const uint32_t i = base + threadIdx.x;
        uint32_t firstPixel = 0, firstLane = 0;
        bool valid = i < n;
        if constexpr (kFirst) {
            const uint32_t start = (((base / kBlock) * fb.laneCount + fb.laneIndex) * kShards + shard) * kBlock;  // frame tile -> first population index
            firstLane = start / fb.plane;                                         // wave-uniform
            firstPixel = start - firstLane * fb.plane + threadIdx.x;
            valid = firstPixel < fb.numPixels;
        }

        // ---- 1. closest hit + surfel (pathTraceKernel :121-163) -----------------------------------
        RayRegs ray;
        ray.o = ray.d = ray.L0 = ray.T = v3(0, 0, 0);
        ray.pix = 0;
        ray.active = false;
        if constexpr (kFirst) {
            if (valid) {
                const uint32_t pixel = firstPixel;
                const PixelCoord pc = locate(tile, pixel);
                loadHome(fb.rngHome, firstLane * fb.plane + pixel, ray.rng);
                const float jitteredX = pc.x + ptrng::uniform(ray.rng);
                const float jitteredY = pc.gy + ptrng::uniform(ray.rng);
                const vec3 start = v3(((jitteredX * eye.invW) - 0.5f) * eye.s,
                                      1 * ((jitteredY * eye.invH) - 0.5f) * eye.s * eye.aspect, 1.0f) *
                                   eye.camera.zNear;
                ray.o = eye.camera.position;
                ray.d = normalize(rotate(eye.camera.rotation, start));
                ray.L0 = v3(0, 0, 0);
                ray.T = v3(1, 1, 1);
                ray.pix = pixel | (firstLane << kLaneShift);
                ray.active = true;
            }
        } else {
            // a ray's planes are fetched where the tile first needs them (origin/direction here, the RNG state before the light
            // samples, radiance/throughput/pixel before the update): 13 fewer live registers across the closest-hit loops
            if (valid) loadRayGeometry<kCoherentIo>(tileBlock(in, base), threadIdx.x, ray);
        }
#if PTSS_ABLATE & 2
        Hit h;
        h.kind = 2; h.idx = (int)(pixOf(ray.pix) % (uint32_t)L.numTriangles); h.distance = 1.0f + ray.d.x;
        h.w0 = 0.3f; h.w1 = 0.3f; h.w2 = 0.4f;
#else
        const Hit h = closestHit<kFirst, kAccel, kBounded>(sc, sceneBlob, L, ray.o, ray.d, valid, reinterpret_cast<uint32_t*>(wq));
#endif
        const bool hit = valid && h.kind != 0;
        if constexpr (!kFirst) {
            if (valid) loadRayRng<kCoherentIo>(tileBlock(in, base), threadIdx.x, ray);
        }
        vec3 point = v3(0, 0, 0), normal = v3(0, 0, 0);
        float cosI = 0;
        int materialIdx = 0;
        if (hit) {
            point = ray.o + ray.d * h.distance;  // Primitives.h:74, :100
            if (h.kind == 1) {
                normal = normalize(point - xyz(loadRow16(sc + L.offSphere + h.idx)));
                materialIdx = reinterpret_cast<const int*>((kAccel ? sceneBlob : sc) + L.offSphereMat)[h.idx];
            } else {
                const float4* nn = sc + L.offTriNormal + 3 * h.idx;
                normal = (xyz(loadRow16(nn)) * h.w0 + xyz(loadRow16(nn + 1)) * h.w1) + xyz(loadRow16(nn + 2)) * h.w2;
                materialIdx = (int)asU(sc[L.offTri + 3 * h.idx].w);
            }
            cosI = dot(-ray.d, normal);
        }
        const bool inside = cosI <= 0.0f;
        const bool lit = hit && !inside && !(PTSS_ABLATE & 1);  // shade() runs, :166-169
        const float4* mat = sc + L.offMaterial + 5 * materialIdx;

        // ---- 2. shade(), CudaTracer.cu:345-390: kNeeLights lights per round through the wave queue ----
        vec3 radiance = v3(0, 0, 0);
        for (int l0 = 0; l0 < numLights; l0 += kNeeLights) {
            uint32_t queued = 0;  // wave-uniform
            bool need[kNeeLights];
            float cosL[kNeeLights], distance2[kNeeLights];
            [[maybe_unused]] vec3 pairW[kNeeLights];      // kPairs: the round's segments stay in registers until both are known
            [[maybe_unused]] float pairReach[kNeeLights];
#pragma unroll
            for (int k = 0; k < kNeeLights; ++k) {
                need[k] = false;
                cosL[k] = 0;
                distance2[k] = 0;
                pairW[k] = v3(0, 0, 0);
                pairReach[k] = 0;
                const int li = l0 + k;
                if (li >= numLights) continue;  // uniform
                vec3 lo = v3(0, 0, 0), w_i = v3(0, 0, 0);
                float distance = 0;
                if (lit) {
                    vec3 lightPoint;
                    if (li < L.numPointLights) {
                        lightPoint = xyz(loadRow16(sc + L.offPointLight + 2 * li));
                    } else {  // getAreaLightPoint :392-418 — four draws whether or not the light ends up visible
                        const float4 light = sc[L.offAreaLight + 2 * (li - L.numPointLights)];
                        const float u1 = ptrng::uniform(ray.rng);
                        const float u2 = ptrng::uniform(ray.rng);
                        const float u3 = ptrng::uniform(ray.rng);
                        const float inverseTotal = ptm::rcp(u1 + u2 + u3);  // 1 / (u1+u2+u3), :403
                        const float weight0 = u1 * inverseTotal, weight1 = u2 * inverseTotal, weight2 = u3 * inverseTotal;
                        // triangleIdx or triangleIdx + 1, :408 — as stored positions (the triangles may be stored grouped by class)
                        const int tri = (ptrng::uniform(ray.rng) > .5f) ? (int)asU(light.w) : (int)asU(sc[L.offAreaLight + 2 * (li - L.numPointLights) + 1].x);
                        const vec3 a = xyz(loadRow16(sc + L.offTri + 3 * tri));
                        const vec3 b = xyz(loadRow16(sc + L.offTriVert + 2 * tri));
                        const vec3 c = xyz(loadRow16(sc + L.offTriVert + 2 * tri + 1));
                        lightPoint = (a * weight0 + b * weight1) + c * weight2;
                    }
                    // head of lineOfSight :423-432
                    const vec3 offset = lightPoint - point;
                    distance2[k] = dot(offset, offset);
                    distance = ptm::sqrt(distance2[k]);
                    w_i = offset / distance;
                    cosL[k] = ptm::max(0.0f, dot(normal, w_i));
                    const bool zeroTerm = L.neeSkipSafe && (distance2[k] > 0.0f) && (distance2[k] < ptm::inf()) &&
                                          (cosL[k] == 0.0f || mat[0].w == 0.0f);
                    need[k] = !zeroTerm;
                    lo = point + (ptm::kRayBump * normal);
                    distance -= 2 * ptm::kRayBump;
                }
                if constexpr (kPairs) {
                    pairW[k] = w_i;
                    pairReach[k] = distance;
                    continue;
                }
                const unsigned long long m = __ballot(need[k]);
                if (need[k]) {
                    const uint32_t slot = queued + __popcll(m & ((1ull << lane) - 1ull));
                    wq[0 * kQueueCap + slot] = lo.x;
                    wq[1 * kQueueCap + slot] = lo.y;
                    wq[2 * kQueueCap + slot] = lo.z;
                    wq[3 * kQueueCap + slot] = w_i.x;
                    wq[4 * kQueueCap + slot] = w_i.y;
                    wq[5 * kQueueCap + slot] = w_i.z;
                    wq[6 * kQueueCap + slot] = distance;
                    wqOwner[slot] = lane | ((uint32_t)k << 8);
                    wqAnswer[k * 64 + lane] = 0;
                }
                queued += (uint32_t)__popcll(m);
            }
            PTSS_DIAG_PAIRS(need[0], need[1], lit);
            waveLdsFence();
            PTSS_DIAG_QUEUE(queued);
            if constexpr (kPairs) {
                // One entry per lane that needs either segment: origin, the two directions and reaches, owner lane | need bits << 8
                // (12 planes of 64). A pass is sized by what it holds: 49+ entries one lane each, fewer -> 32 / 16 / 8 entries
                // with 2 / 4 / 8 lanes sharing the primitive list.
                constexpr int kPairCap = 64;
                const bool needAny = need[0] || need[1];
                const unsigned long long m = __ballot(needAny);
                const uint32_t pairs = (uint32_t)__popcll(m);
                uint32_t* pairOwner = reinterpret_cast<uint32_t*>(wq + 11 * kPairCap);
                if (needAny) {
                    const uint32_t slot = __popcll(m & ((1ull << lane) - 1ull));
                    const vec3 lo = point + (ptm::kRayBump * normal);
                    wq[0 * kPairCap + slot] = lo.x;
                    wq[1 * kPairCap + slot] = lo.y;
                    wq[2 * kPairCap + slot] = lo.z;
                    wq[3 * kPairCap + slot] = pairW[0].x;
                    wq[4 * kPairCap + slot] = pairW[0].y;
                    wq[5 * kPairCap + slot] = pairW[0].z;
                    wq[6 * kPairCap + slot] = pairReach[0];
                    wq[7 * kPairCap + slot] = pairW[1].x;
                    wq[8 * kPairCap + slot] = pairW[1].y;
                    wq[9 * kPairCap + slot] = pairW[1].z;
                    wq[10 * kPairCap + slot] = pairReach[1];
                    pairOwner[slot] = lane | ((need[0] ? 1u : 0u) << 8) | ((need[1] ? 1u : 0u) << 9);
                    wqAnswer[0 * 64 + lane] = 0;
                    wqAnswer[1 * 64 + lane] = 0;
                }
                waveLdsFence();
                for (uint32_t e0 = 0; e0 < pairs;) {
                    const uint32_t rem = pairs - e0;
                    const uint32_t units = (rem + 7u) >> 3;
                    const int chunkLog = (units >= 7u) ? 6 : (units >= 4u ? 5 : (units >= 2u ? 4 : 3));
                    const int shift = 6 - chunkLog;
                    const uint32_t mine = lane >> shift;
                    const uint32_t sub = lane & ((1u << shift) - 1u);
                    const bool have = mine < rem;
                    const uint32_t es = have ? e0 + mine : 0u;
                    const vec3 lo = v3(wq[0 * kPairCap + es], wq[1 * kPairCap + es], wq[2 * kPairCap + es]);
                    const vec3 wA = v3(wq[3 * kPairCap + es], wq[4 * kPairCap + es], wq[5 * kPairCap + es]);
                    const float dA = wq[6 * kPairCap + es];
                    const vec3 wB = v3(wq[7 * kPairCap + es], wq[8 * kPairCap + es], wq[9 * kPairCap + es]);
                    const float dB = wq[10 * kPairCap + es];
                    const uint32_t ow = pairOwner[es];
                    const bool liveA = have && ((ow >> 8) & 1u) != 0u, liveB = have && ((ow >> 9) & 1u) != 0u;
                    bool occA, occB;
                    if (shift == 0) pairAnyHit<kBounded, false>(sc, L, lo, wA, dA, liveA, wB, dB, liveB, 0, 0, occA, occB);
                    else pairAnyHit<kBounded, true>(sc, L, lo, wA, dA, liveA, wB, dB, liveB, shift, (int)sub, occA, occB);
                    const unsigned long long verdictsA = __ballot(occA), verdictsB = __ballot(occB);
                    const unsigned long long group = ((1ull << (1u << shift)) - 1ull) << (mine << shift);
                    if (have && sub == 0u) {
                        if ((verdictsA & group) != 0ull) wqAnswer[0 * 64 + (ow & 63u)] = 1;
                        if ((verdictsB & group) != 0ull) wqAnswer[1 * 64 + (ow & 63u)] = 1;
                    }
                    e0 += 1u << chunkLog;
                }
            } else {
            // Passes over the wave's queue, each sized by what is left (wave-uniform): 49+ segments -> a dense pass,
            // one lane per segment (anyHit, broadcast rows); fewer -> a chunk of 32 / 16 / 8 segments with 2 / 4 / 8
            // lanes per segment sharing the primitive list (anyHitSplit), so that a pass costs about what it holds:
            // 77 segments = 1 + 1/4 dense passes instead of 2, 13 segments = 1/4 instead of 1.
            for (uint32_t e0 = 0; e0 < queued;) {
                const uint32_t rem = queued - e0;
                const uint32_t units = (rem + 7u) >> 3;  // of 8 segments
                // (the chunked sphere traversal is per lane already: those scenes take dense passes only)
                const int chunkLog = (units >= 7u || kAccel) ? 6 : (units >= 4u ? 5 : (units >= 2u ? 4 : 3));
                const int shift = 6 - chunkLog;                      // lanes per segment = 1 << shift
                const uint32_t mine = lane >> shift;                 // this lane's segment within the chunk
                const uint32_t sub = lane & ((1u << shift) - 1u);    // its share of the primitive list
                const bool have = mine < rem;
                const uint32_t es = have ? e0 + mine : 0u;
                const vec3 lo = v3(wq[0 * kQueueCap + es], wq[1 * kQueueCap + es], wq[2 * kQueueCap + es]);
                const vec3 wi = v3(wq[3 * kQueueCap + es], wq[4 * kQueueCap + es], wq[5 * kQueueCap + es]);
                const float reach = wq[6 * kQueueCap + es];
                bool occ;
                if (kAccel && (e0 != 0u || queued <= 64u)) {  // dense pass whose OTHER half of the queue planes is free
                    occ = anySpheresHybrid(sc, L, wq + e0, wq + (e0 == 0u ? 64 : 0), lo, wi, reach, have);
                    occ = occ || anyTriangles(sc, L, lo, wi, reach, have && !occ);
                } else {
                    occ = (shift == 0) ? anyHit<kAccel, kBounded>(sc, L, lo, wi, reach, have) : anyHitSplit<kBounded>(sc, L, lo, wi, reach, have, shift, (int)sub);
                }
                const unsigned long long verdicts = __ballot(occ);  // all lanes vote before anyone branches
                const unsigned long long group = ((1ull << (1u << shift)) - 1ull) << (mine << shift);
                if (have && sub == 0u && (verdicts & group) != 0ull) {
                    const uint32_t ow = wqOwner[es];
                    wqAnswer[(ow >> 8) * 64 + (ow & 63u)] = 1;
                }
                e0 += 1u << chunkLog;
            }
            }
            waveLdsFence();
#pragma unroll
            for (int k = 0; k < kNeeLights; ++k) {
                const int li = l0 + k;
                if (li >= numLights) continue;
                if (need[k] && wqAnswer[k * 64 + lane] == 0) {
                    const vec3 power = (li < L.numPointLights) ? xyz(loadRow16(sc + L.offPointLight + 2 * li + 1))
                                                               : xyz(loadRow16(sc + L.offAreaLight + 2 * (li - L.numPointLights)));
                    addLambertTerm(radiance, cosL[k], power, distance2[k], mat[0]);
                }
            }
            waveLdsFence();
        }

        // ---- 3. scatter + radiance update (pathTraceKernel :172-198) -------------------------------
        bool alive = false;
        if constexpr (!kFirst) {
            if (valid) loadRayRadiance<kCoherentIo>(tileBlock(in, base), threadIdx.x, ray);
        }
        if (valid) {
            if (hit) {
                // emmitance, :163 — read here, after the shadow passes, instead of being held in registers across them
                vec3 directRadiance = v3(0, 0, 0) + xyz(mat[3]);
                if (lit) directRadiance = directRadiance + radiance;
                vec3 indirectRadiance = v3(1, 1, 1);
                if (!kLast && !(PTSS_ABLATE & 4)) indirectRadiance = scatter(mat, ray, point, normal, cosI);
                if (inside) {  // Beer-Lambert, :179-185
                    const float4 ab = mat[2];
                    ray.T = ray.T * v3(ptm::exp(-h.distance * ab.x), ptm::exp(-h.distance * ab.y),
                                       ptm::exp(-h.distance * ab.z));
                }
                ray.L0 = ray.L0 + ray.T * directRadiance;
                ray.T = ray.T * indirectRadiance;
            } else {  // :193-198
                const vec3 dc = v3(fb.defaultColor[0], fb.defaultColor[1], fb.defaultColor[2]);
                ray.L0 = ray.L0 + dc * ray.T;
                ray.active = false;
            }
            alive = ray.active && !kLast;
        }

        // ---- 4+5. stream compaction of the survivors (replaces thrust::partition, :629) and
        // writeToPixelsKernel for the paths that ended. Each wave compacts on its own — 64-bit ballot, popcount lane rank, ONE
        // returning atomic per wave on the shard's counter (16 counters share the load) — no barrier; the atomic is issued
        // first so that its round trip hides behind the tone-mapping of the finished lanes.
        uint32_t slot = 0;
        if constexpr (!kLast) {
            const unsigned long long live = __ballot(alive);
            if (live) {
                const int leader = __ffsll((long long)live) - 1;
                uint32_t base0 = 0;
                if ((int)lane == leader)
                    base0 = atomicAdd(&fb.counts[countIndex(bounce + 1, (int)shard)], (uint32_t)__popcll(live));
                slot = base0;  // consumed after the finish work below
                if (valid && !alive && !(PTSS_ABLATE & 8)) finishPath(fb, ray, quantT);
                slot = __shfl(slot, leader) + __popcll(live & ((1ull << lane) - 1ull));
                if (alive) storeRay<kCoherentIo>(out, slot, ray);
            } else if (valid && !(PTSS_ABLATE & 8)) {
                finishPath(fb, ray, quantT);
            }
        } else {
            if (valid && !(PTSS_ABLATE & 8)) finishPath(fb, ray, quantT);
        }
    }
}

template <bool kLast, bool kSceneInLds, bool kFirst, bool kAccel, bool kBounded, bool kPairsWanted>
__device__ __forceinline__ void bounceBody(const FrameBuffers& fb, const float4* __restrict__ sceneBlob, const SceneLayout& L, int bounce,
                                           const TileMap& tile, const EyeParams& eye) {
    extern __shared__ __attribute__((aligned(256))) float4 lds[];
    constexpr bool kPairs = kPairsWanted && !kAccel;   // the two shadow segments of a surface point travel and are tested together (SceneLayout::neePairs)
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = fb.counts[countIndex(bounce, (int)shard)];  // this shard's live rays (of this lane)
    if constexpr (kFirst) {
        if (fb.frameRays <= fb.minLive) return;  // loop guard, CudaTracer.cu:622: the frame starts with <= 128 rays
    } else {
        if (n == 0) return;  // nothing of this shard reached this bounce (whatever the guard says)
        if (n <= fb.minLive) {  // loop guard on the FRAME's live count (device-side; every workgroup that has work reaches
            uint32_t own = 0;   // the same verdict). Only a nearly empty shard has to add up; only a nearly empty LANE asks its peers.
            for (int s = 0; s < kShards; ++s) own += fb.counts[countIndex(bounce, s)];
            if (own <= fb.minLive && frameLiveCount(fb, bounce, own, fb.peerTarget) <= fb.minLive) return;
        }
    }

    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    float4* work = lds + (kSceneInLds ? L.ldsVec4 : 0);
    float* wq = reinterpret_cast<float*>(work + kBlockScratchVec4) + wave * kWaveLdsWords;  // this wave's queue
    uint32_t* wqOwner = reinterpret_cast<uint32_t*>(wq + 7 * kQueueCap);
    unsigned char* wqAnswer = reinterpret_cast<unsigned char*>(wqOwner + kQueueCap);  // [kNeeLights][64], 0 / 1

    const float4* sc;
    if constexpr (kSceneInLds) {
        for (int k = threadIdx.x; k < L.ldsVec4; k += kBlock) lds[k] = sceneBlob[k];
        __syncthreads();
        sc = lds;
    } else {
        sc = sceneBlob;
    }
    const float* quantT = reinterpret_cast<const float*>(sc + L.offQuant);

    const size_t regionWords = (size_t)fb.regionCap * kRayPlanes;
    const float* __restrict__ in = fb.pool[bounce & 1] + shard * regionWords;  // this shard's region
    float* __restrict__ out = fb.pool[(bounce + 1) & 1] + shard * regionWords;
    const int numLights = L.numPointLights + L.numAreaLights;
    const TileEnv env{sc, sceneBlob, quantT, in, out, wq, wqOwner, wqAnswer, shard, lane, n, numLights, bounce};

    // one tile per workgroup when the host's grid hint is right; grid-stride keeps any n correct
    // bounce 0 walks the FRAME's tiles (S sample planes of fb.plane pixels; tile t belongs to shard t % kShards),
    // every later bounce walks the shard's compacted region
    // (with frame lanes: round R of the frame belongs to lane R % laneCount, whose round R / laneCount it is)
    const uint32_t roundsOfShard = (fb.firstTiles + kShards - 1 - shard) / kShards;
    const uint32_t span = kFirst ? ((roundsOfShard + fb.laneCount - 1 - fb.laneIndex) / fb.laneCount) * kBlock : n;
    for (uint32_t base = (blockIdx.x / kShards) * kBlock; base < span; base += (gridDim.x / kShards) * kBlock) {
        bounceTile<kLast, kFirst, kAccel, kBounded, kPairs, false>(fb, L, tile, eye, env, base);
    }
}

template <bool kLast, bool kSceneInLds, bool kFirst, bool kAccel, bool kBounded, bool kPairs>
__global__ __launch_bounds__(kBlock, kAccel ? 4 : (kFirst ? PTSS_MINWAVES_FIRST : (kBounded ? PTSS_MINWAVES_BOUNDED : PTSS_MINWAVES))) void bounceKernel(  // chunked scenes: their LDS image (21 KB + the work area) admits four workgroups per CU, so four waves per SIMD = 128 registers cost nothing (round 3: 5 -> 4, no scratch, c5 +2.8 %)
    FrameBuffers fb, const float4* __restrict__ sceneBlob, SceneLayout L, int bounce, TileMap tile, EyeParams eye) {
    bounceBody<kLast, kSceneInLds, kFirst, kAccel, kBounded, kPairs>(fb, sceneBlob, L, bounce, tile, eye);
    // frame lanes: "this workgroup of bounce `bounce` has ended" (every workgroup, also one that had nothing to do) — the
    // peers' loop guard of bounce + 1 waits for the whole grid (frameLiveCount). The survivor counters were raised by
    // returning device-scope atomics, so they have been performed when a wave gets here, and the barrier collects the
    // workgroup's waves: a relaxed add is enough. (A RELEASE here writes the XCD's L2 back once per workgroup: 3x slower.)
    if (fb.numPeers != 0) {
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(fb.myDone + countIndex(bounce, (int)(blockIdx.x % kShards)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- ONE LAUNCH PER FRAME (frames whose bounce-0 tiles are all resident at once: up to ~3 * 2^17 rays per pass) -------------
// A pass of a small frame is up to 16 launches that are each at most one resident round wide: such a launch costs what ONE
// tile costs from end to end (~15 us at 512 x 512: scene staging, ray fetch, ~3,300 dependent instructions, the counter
// round trip) whatever the machine could do meanwhile. Here workgroup w = (shard s = w % kShards, tile j =
// w / kShards) stays: it stages the scene once, traces bounce-0 tile j of its shard, and then, bounce after bounce, tile j of
// the shard's compacted region for as long as that tile exists (a region never grows, so a workgroup that finds its tile
// beyond the count is done for good).
// Hand-off between the bounces of ONE shard (the only dependency: survivors are compacted per shard): every workgroup that
// traced a tile of bounce b adds 1 to done[b][s] — word kDoneWord of the (b, s) counter line — after its waves have drained
// their stores (s_waitcnt vmcnt(0), workgroup barrier); a workgroup goes on to bounce b + 1 when done[b][s] has reached the
// tiles the shard had at bounce b (it knows: ceil(n_s(b) / 256)), and then reads n_s(b + 1), raised by returning atomics
// before those arrivals. The rays themselves cross between workgroups through sc1 accesses (ldPlane<true> / stPlane<true>):
// written through, read past the L1 — MI355X_MICROARCH.md's hand-off "one lane of each storing workgroup signals by an
// agent-scope atomic add, the consumer polls that counter with sc1 loads, payload stored and loaded sc1" (no L2 write-back,
// no L1 invalidate per bounce). The loop guard `numRays > 128` (CudaTracer.cu:622) is a whole-frame count: a workgroup whose
// shard still holds more than 128 rays knows the frame does; otherwise lanes 0..15 of its first wave each follow one shard's
// chain of done counters up to this bounce (a few loads, progress kept in registers) and add the shards' counts up.
// Deadlock freedom: the host launches this kernel only when the whole grid is resident at once (ptss_api.hip), so every
// workgroup a waiter depends on is running or has finished; every wait is bounded all the same (peerWaitExpired) and a
// workgroup whose wait expires leaves — the host then reports PTSS_ETIMEOUT.
constexpr int kDoneWord = 1;

__device__ __forceinline__ bool waitForCount(const uint32_t* word, uint32_t target) {   // bounded; true = reached
    unsigned long long since = 0ull;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (peerWaitExpired(since)) return false;
    }
    return true;
}

template <bool kAccel, bool kBounded, bool kPairsWanted>
__global__ __launch_bounds__(kBlock, kAccel ? 4 : (kBounded ? PTSS_MINWAVES_BOUNDED : PTSS_MINWAVES)) void frameKernel(
    FrameBuffers fb, const float4* __restrict__ sceneBlob, SceneLayout L, int numBounces, TileMap tile, EyeParams eye) {
    extern __shared__ __attribute__((aligned(256))) float4 lds[];
    constexpr bool kPairs = kPairsWanted && !kAccel;
    if (fb.frameRays <= fb.minLive) return;   // loop guard at bounce 0: the frame starts with <= 128 rays
    const uint32_t shard = blockIdx.x % kShards, myTile = blockIdx.x / kShards;
    const uint32_t rounds = (fb.firstTiles + kShards - 1 - shard) / kShards;   // bounce-0 tiles of this shard
    if (myTile >= rounds) return;

    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    float4* work = lds + L.ldsVec4;
    uint32_t* bcast = reinterpret_cast<uint32_t*>(work);   // [0] rays of this shard entering the next bounce (~0u: give up), [1] the frame's
    float* wq = reinterpret_cast<float*>(work + kBlockScratchVec4) + wave * kWaveLdsWords;
    uint32_t* wqOwner = reinterpret_cast<uint32_t*>(wq + 7 * kQueueCap);
    unsigned char* wqAnswer = reinterpret_cast<unsigned char*>(wqOwner + kQueueCap);
    for (int k = threadIdx.x; k < L.ldsVec4; k += kBlock) lds[k] = sceneBlob[k];   // the scene, once per frame
    __syncthreads();
    const size_t regionWords = (size_t)fb.regionCap * kRayPlanes;
    TileEnv env{lds, sceneBlob, reinterpret_cast<const float*>(lds + L.offQuant), nullptr, nullptr, wq, wqOwner, wqAnswer, shard, lane, 0u,
                L.numPointLights + L.numAreaLights, 0};
    uint32_t tilesNow = rounds, raysNow = 0;
    // the guard's view of the other shards (lanes 0..15 of wave 0, one shard each): the next bounce to verify, -1 = that shard is empty
    int verified = 0;
    for (int b = 0; b < numBounces; ++b) {
        const bool last = b == numBounces - 1;
        env.bounce = b;
        env.n = raysNow;
        env.in = fb.pool[b & 1] + shard * regionWords;
        env.out = fb.pool[(b + 1) & 1] + shard * regionWords;
        const uint32_t base = myTile * kBlock;
        if (b == 0) {
            if (last) bounceTile<true, true, kAccel, kBounded, kPairs, true>(fb, L, tile, eye, env, base);
            else bounceTile<false, true, kAccel, kBounded, kPairs, true>(fb, L, tile, eye, env, base);
        } else if (last) {
            bounceTile<true, false, kAccel, kBounded, kPairs, true>(fb, L, tile, eye, env, base);
        } else {
            bounceTile<false, false, kAccel, kBounded, kPairs, true>(fb, L, tile, eye, env, base);
        }
        if (last) break;
        // publish this tile (its survivors are in the output region, the shard's next count was raised by returning atomics) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(fb.counts + countIndex(b, (int)shard) + kDoneWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... and wait for the shard's other tiles of this bounce; n_s(b + 1) is final once those arrivals are complete
            const uint32_t* done = fb.counts + countIndex(b, (int)shard) + kDoneWord;
            uint32_t next = ~0u;
            unsigned long long since = 0ull;
            for (;;) {
                if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tilesNow) {
                    next = __hip_atomic_load(fb.counts + countIndex(b + 1, (int)shard), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (peerWaitExpired(since)) {
                    atomicAdd(fb.guardTimeouts, 1u);
                    break;
                }
            }
            bcast[0] = next;
        }
        __syncthreads();
        const uint32_t next = bcast[0];
        if (next == ~0u) return;
        if (myTile * kBlock >= next) return;   // no tile of bounce b + 1 for this workgroup, nor of any later bounce
        if (next <= fb.minLive) {   // the loop guard is the FRAME's count: ask the other shards (wave 0, one lane per shard)
            if (wave == 0) {
                uint32_t theirs = 0;
                bool ok = true;
                if (lane < kShards) {
                    const int s = (int)lane;
                    while (verified >= 0 && verified <= b) {   // every bounce up to b of shard s must have ended before its count of b + 1 is final
                        const uint32_t tiles = verified == 0 ? (fb.firstTiles + kShards - 1 - (uint32_t)s) / kShards
                                                             : (__hip_atomic_load(fb.counts + countIndex(verified, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + kBlock - 1) / kBlock;
                        if (tiles == 0) {
                            verified = -1;   // nothing of shard s reached this bounce: nothing ever will
                        } else if (waitForCount(fb.counts + countIndex(verified, s) + kDoneWord, tiles)) {
                            ++verified;
                        } else {
                            ok = false;
                            break;
                        }
                    }
                    if (verified > b) theirs = __hip_atomic_load(fb.counts + countIndex(b + 1, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t total = theirs;
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) total += (uint32_t)__shfl_xor((int)total, off);
                const bool allOk = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
                if (lane == 0) {
                    if (!allOk) atomicAdd(fb.guardTimeouts, 1u);
                    bcast[1] = allOk ? total : 0u;   // a wait that expired: leave (as if the guard had stopped the frame)
                }
            }
            __syncthreads();
            if (bcast[1] <= fb.minLive) return;
        }
        tilesNow = (next + kBlock - 1) / kBlock;
        raysNow = next;
    }
}

// After the last launched bounce: tone-map whatever the loop guard left alive (<= 128 rays in all
// shards together), and add this frame's ray-bounce total to the running counter.
__global__ void flushKernel(FrameBuffers fb, int numBounces, FlushTargets targets) {
    __shared__ uint32_t totals[kMaxBounces + 1];
    __shared__ int stopShared;
    for (int b = threadIdx.x; b <= numBounces; b += blockDim.x) {
        uint32_t total = 0;
        for (int s = 0; s < kShards; ++s) total += fb.counts[countIndex(b, s)];
        totals[b] = total;  // this lane's rays entering bounce b
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // where did the frame's loop guard stop? (the bounce kernels decided the same way)
        int stop = numBounces;
        unsigned long long sum = 0;
        for (int b = 0; b < numBounces; ++b) {
            uint32_t frame = totals[b];
            if (frame <= fb.minLive) {
                if (b == 0) {
                    frame = fb.frameRays;
                } else {
                    uint32_t want[kMaxLanes - 1];
                    for (int p = 0; p < kMaxLanes - 1; ++p) want[p] = targets.target[p][b - 1];
                    frame = frameLiveCount(fb, b, frame, want);
                }
            }
            if (frame <= fb.minLive) {
                stop = b;
                break;
            }
            sum += totals[b];
        }
        *fb.totalRayBounces += sum;  // this lane's slot
        stopShared = stop;
    }
    __syncthreads();
    const int stop = stopShared;
    const uint32_t i = threadIdx.x;
    if (stop == 0) {
        // Not even bounce 0 ran (<= 128 rays in the frame). Eye rays are made inside bounce 0, so there are
        // none in the pool: do what computeEyeRaysKernel + writeToPixelsKernel would have done to each pixel —
        // two jitter draws, then a sample of radiance 0. (Lane 0 does it for the whole frame.)
        if (fb.laneIndex == 0 && i < fb.numPixels)
            for (uint32_t l = 0; l < fb.samples; ++l) {
                RayRegs ray;
                loadHome(fb.rngHome, l * fb.plane + i, ray.rng);
                (void)ptrng::uniform(ray.rng);
                (void)ptrng::uniform(ray.rng);
                ray.L0 = v3(0, 0, 0);
                ray.pix = i | (l << kLaneShift);
                finishPath(fb, ray, fb.quantTable);
            }
    } else if (stop < numBounces && totals[stop] != 0) {  // the guard left this lane's rays alive: writeToPixelsKernel for them
        for (int s = 0; s < kShards; ++s) {
            // n <= 128 = blockDim.x here by the guard that stopped the loop; clamped all the same, and a slot whose pixel
            // or sample lane is out of range is skipped rather than written through: a stale slot must never be able to
            // fault the device (a GPU memory fault aborts the calling process — DESIGN.md §4a, the round-1 abort)
            uint32_t n = fb.counts[countIndex(stop, s)];
            n = n < blockDim.x ? n : blockDim.x;
            if (i < n) {
                RayRegs ray;
                loadRay(tileBlock(fb.pool[stop & 1] + (size_t)s * fb.regionCap * kRayPlanes, (i / kBlock) * kBlock), i % kBlock, ray);
                if (pixOf(ray.pix) < fb.numPixels && laneOf(ray.pix) < fb.samples) finishPath(fb, ray, fb.quantTable);
            }
        }
    }
    // keep this frame's counters for the host (live counts, grid hints) and arm the OTHER buffer for the next frame:
    // bounce 0 = the shard's pixels, every later bounce 0. This frame's buffer stays as it is: a peer lane may still read it.
    // The other buffer is the one of the frame before this: a peer that runs a frame behind may still be reading it, so
    // wait until every peer has finished that frame (its flushKernel was enqueued before this one: no deadlock in a shared queue).
    if (threadIdx.x == 0)
        for (uint32_t p = 0; p < fb.numPeers; ++p) {
            unsigned long long since = 0ull;
            while ((int32_t)(__hip_atomic_load(fb.peerFrameDone[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fb.frameSeq) < 0) {
                __builtin_amdgcn_s_sleep(64);
                if (peerWaitExpired(since)) {
                    atomicAdd(fb.guardTimeouts, 1u);
                    break;
                }
            }
        }
    __syncthreads();
    // (every bounce slot, not just this frame's: the other buffer still holds the counts of two frames ago, and the
    // bounce count may change between frames — ptss_set_mode, ptss_set_max_iterations)
    for (int k = threadIdx.x; k < (kMaxBounces + 1) * kShards; k += blockDim.x) {
        const int b = k / kShards, s = k % kShards;
        fb.lastCounts[countIndex(b, s)] = (b <= numBounces) ? fb.counts[countIndex(b, s)] : 0u;
        fb.countsNext[countIndex(b, s)] = (b == 0) ? fb.shardCount0[s] : 0u;
        fb.countsNext[countIndex(b, s) + kDoneWord] = 0u;   // frameKernel's finished-tiles counter of that line
    }
    // This lane has finished the frame: every read of a peer's counters (thread 0, above) has returned by now, and what this
    // kernel wrote — finishPath's accumulator / pixel / float-sum / RNG words of the leftover rays, lastCounts, countsNext, the
    // ray-bounce total — is PUBLISHED before the frame-done word: every wave drains its stores, the workgroup meets, and one
    // lane writes this XCD's L2 back (agent-scope release; once per frame and lane, in a 128-thread kernel) before it stores
    // the word. The join below depends on that: the caller's stream is ordered behind the LAST lane's flush only, and the
    // end-of-kernel write-back of that kernel covers its own XCD's L2, not the one a peer's flush ran on.
    if (fb.numPeers != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fb.joinsFrame) {  // the last lane's flush ends only when the whole frame has (the join event sits behind it)
                for (uint32_t p = 0; p < fb.numPeers; ++p) {
                    unsigned long long since = 0ull;
                    while ((int32_t)(__hip_atomic_load(fb.peerFrameDone[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (fb.frameSeq + 1u)) < 0) {
                        __builtin_amdgcn_s_sleep(64);
                        if (peerWaitExpired(since)) {
                            atomicAdd(fb.guardTimeouts, 1u);
                            break;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // one poll loop per peer, then ONE acquire
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-back has completed before the word goes out (hipcc may drop the fence's own wait)
            __hip_atomic_store(fb.myFrameDone, fb.frameSeq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// =================================================================================================
static inline unsigned blocksFor(uint32_t n, unsigned block) { return (n + block - 1) / block; }

hipError_t launchRngInit(hipStream_t st, uint32_t* rngHome, uint32_t plane, uint32_t samples, TileMap tile, uint64_t seed,
                         const uint32_t* jumpTable) {
    hipLaunchKernelGGL(rngInitKernel, dim3(blocksFor(plane * samples, 256)), dim3(256), 0, st, rngHome, plane, samples, tile, seed,
                       jumpTable);
    return hipGetLastError();
}

hipError_t launchDisplay(hipStream_t st, const FrameBuffers& fb) {
    hipLaunchKernelGGL(displayKernel, dim3(blocksFor(fb.numPixels, 256)), dim3(256), 0, st, fb);
    return hipGetLastError();
}

hipError_t launchClear(hipStream_t st, const FrameBuffers& fb) {
    hipLaunchKernelGGL(clearKernel, dim3(blocksFor(fb.numPixels, 256)), dim3(256), 0, st, fb);
    return hipGetLastError();
}

hipError_t launchPrimaryPrep(hipStream_t st, float4* sceneBlob, const SceneLayout& layout, ptss_vec3 origin) {
    const int n = layout.numSpheres > layout.numTriangles ? layout.numSpheres : layout.numTriangles;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(primaryPrepKernel, dim3(blocksFor((uint32_t)n, 64)), dim3(64), 0, st, sceneBlob, layout, origin);
    return hipGetLastError();
}

template <bool kLast, bool kLds, bool kFirst, bool kAccel, bool kBounded, bool kPairs>
static hipError_t launchBounceT(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, const SceneLayout& layout,
                                int bounce, int gridBlocks, const TileMap& tile, const EyeParams& eye) {
    const size_t lds = bounceLdsBytes(layout, kLds);
    hipLaunchKernelGGL((bounceKernel<kLast, kLds, kFirst, kAccel, kBounded, kPairs>), dim3(gridBlocks), dim3(kBlock), lds, st, fb, sceneBlob, layout, bounce,
                       tile, eye);
    return hipGetLastError();
}

#if PTSS_DIAG
hipError_t readDiagCounters(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag), 64); }
#endif
size_t bounceLdsBytes(const SceneLayout& layout, bool sceneInLds) {
    return ((sceneInLds ? (size_t)layout.ldsVec4 : 0) + kBlockLdsVec4) * sizeof(float4);
}

hipError_t launchBounce(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, SceneLayout layout, int bounce,
                        bool isLast, bool sceneInLds, bool bounded, int gridBlocks, TileMap tile, EyeParams eye) {
    const bool isFirst = bounce == 0;
#define PTSS_GO(a, b, c)                                                                                   \
    do {                                                                                                   \
        if (layout.accelSpheres) return launchBounceT<a, b, c, true, false, false>(st, fb, sceneBlob, layout, bounce, gridBlocks, tile, eye); \
        if (bounded && layout.neePairs) return launchBounceT<a, b, c, false, true, true>(st, fb, sceneBlob, layout, bounce, gridBlocks, tile, eye); \
        if (bounded) return launchBounceT<a, b, c, false, true, false>(st, fb, sceneBlob, layout, bounce, gridBlocks, tile, eye);     \
        return launchBounceT<a, b, c, false, false, false>(st, fb, sceneBlob, layout, bounce, gridBlocks, tile, eye);                 \
    } while (0)
    if (sceneInLds) {
        if (isFirst) { if (isLast) PTSS_GO(true, true, true); else PTSS_GO(false, true, true); }
        if (isLast) PTSS_GO(true, true, false); else PTSS_GO(false, true, false);
    }
    if (isFirst) { if (isLast) PTSS_GO(true, false, true); else PTSS_GO(false, false, true); }
    if (isLast) PTSS_GO(true, false, false); else PTSS_GO(false, false, false);
#undef PTSS_GO
}

template <bool kAccel, bool kBounded, bool kPairs>
static hipError_t launchFrameT(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, const SceneLayout& layout, int numBounces, int gridBlocks,
                               const TileMap& tile, const EyeParams& eye) {
    hipLaunchKernelGGL((frameKernel<kAccel, kBounded, kPairs>), dim3(gridBlocks), dim3(kBlock), bounceLdsBytes(layout, true), st, fb, sceneBlob, layout,
                       numBounces, tile, eye);
    return hipGetLastError();
}
hipError_t launchFrame(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, SceneLayout layout, int numBounces, bool bounded, int gridBlocks,
                       TileMap tile, EyeParams eye) {
    if (layout.accelSpheres) return launchFrameT<true, false, false>(st, fb, sceneBlob, layout, numBounces, gridBlocks, tile, eye);
    if (bounded && layout.neePairs) return launchFrameT<false, true, true>(st, fb, sceneBlob, layout, numBounces, gridBlocks, tile, eye);
    if (bounded) return launchFrameT<false, true, false>(st, fb, sceneBlob, layout, numBounces, gridBlocks, tile, eye);
    return launchFrameT<false, false, false>(st, fb, sceneBlob, layout, numBounces, gridBlocks, tile, eye);
}
// resident workgroups per CU of the frame kernel `layout` would run (the API's answer; the caller keeps one in reserve)
int frameOccupancyBlocksPerCU(const SceneLayout& layout, bool bounded) {
    const size_t lds = bounceLdsBytes(layout, true);
    int a = 0;
    hipError_t e;
    if (layout.accelSpheres) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, frameKernel<true, false, false>, kBlock, lds);
    else if (bounded && layout.neePairs) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, frameKernel<false, true, true>, kBlock, lds);
    else if (bounded) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, frameKernel<false, true, false>, kBlock, lds);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, frameKernel<false, false, false>, kBlock, lds);
    return e == hipSuccess ? a : 0;
}

hipError_t launchFlush(hipStream_t st, const FrameBuffers& fb, int numBounces, const FlushTargets& targets) {
    hipLaunchKernelGGL(flushKernel, dim3(1), dim3(kMinLiveRays), 0, st, fb, numBounces, targets);
    return hipGetLastError();
}

// Resident workgroups per CU of the mid-bounce instantiation that `layout` runs (registers and this scene's LDS image)
int bounceOccupancyBlocksPerCU(const SceneLayout& layout, bool sceneInLds, bool accel) {
    const size_t lds = bounceLdsBytes(layout, sceneInLds);
    int a = 0;
    hipError_t e;
    if (accel)
        e = sceneInLds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, true, false, true, false, false>, kBlock, lds)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, false, false, true, false, false>, kBlock, lds);
    else if (layout.sphereBounded)   // (the paired-shadow instantiations have the same launch bounds)
        e = sceneInLds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, true, false, false, true, false>, kBlock, lds)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, false, false, false, true, false>, kBlock, lds);
    else
        e = sceneInLds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, true, false, false, false, false>, kBlock, lds)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, false, false, false, false, false>, kBlock, lds);
    return e == hipSuccess ? a : 0;
}

}  // namespace ptss
