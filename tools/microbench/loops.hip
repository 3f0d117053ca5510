// loops.hip — the bounce kernel's inner loops in isolation, at the bounce kernel's occupancy: how many SIMD cycles does
// ONE wave spend per call of the sphere candidate pass / the closest-hit loops / the any-hit loops, against the 2 cycles
// per VALU instruction the SIMD can issue? Includes the product's kernel source, so every -D build switch applies:
//   hipcc <build.py HIP_FLAGS> -I include -I cuda-path-tracer-ss_amd/csrc [-DPTSS_...] tools/microbench/loops.hip -o loops
// Measurement tool only: never built by build.py, never loaded by the product or the tests.
// HISTORICAL: written against round 2's kernel source (git show c7be1ea:cuda-path-tracer-ss_amd/csrc/ptss_kernels.hip). Round 3
// pruned the functions and switches it calls (sphereMayHit, triangleTest, PTSS_SPHERE_UNROLL, PTSS_ROW128) and stores the
// triangles grouped by edge class, so it does not compile against the present source; its numbers are the round-2 ones quoted
// in profiles/README.md and DESIGN.md §6. Round 3's loop measurements are counter-based instead (tools/valu_per_wave.py,
// tools/pmc_counters.py, the ablation builds a1/a2/a3 of tools/build_variants.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#include "../../cuda-path-tracer-ss_amd/csrc/ptss_kernels.hip"

using namespace ptss;

// ---- experimental triangle loops (what costs the 160 cycles per triangle?) ----
__device__ __forceinline__ float rcpFast(float x) {  // no range guard (valid for |x| < 2^126)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
struct Best { float dist; int idx; float w0, w1, w2; };
// straight-line: no wave-uniform exits at all, results merged with selects
template <bool kGuard>
__device__ __forceinline__ void triStraight(const TriRows& tr, int i, vec3 o, vec3 d, Best& best) {
    const vec3 v0 = xyz(tr.a), e1 = xyz(tr.b), e2 = xyz(tr.c);
    const vec3 q = cross(d, e2);
    const float det = dot(e1, q);
    const float inverseDet = kGuard ? triRcp(det) : rcpFast(det);
    const vec3 s = o - v0;
    const vec3 r = cross(s, e1);
    const float dist = dot(e2, r) * inverseDet;
    const float b1 = dot(s, q) * inverseDet;
    const float b2 = dot(d, r) * inverseDet;
    const float b0 = 1.0f - (b1 + b2);
    const unsigned long long hitMask = maskOf(!(ptm::abs(det) <= 1e-7f)) & maskOf(!(dist <= 0.0f)) & maskOf(!(dist > best.dist)) &
                                       maskOf(!(b0 < 0)) & maskOf(!(b1 < 0)) & maskOf(!(b2 < 0));
    const bool hit = __builtin_amdgcn_inverse_ballot_w64(hitMask);
    best.dist = hit ? dist : best.dist;
    best.idx = hit ? i : best.idx;
    best.w0 = hit ? b0 : best.w0;
    best.w1 = hit ? b1 : best.w1;
    best.w2 = hit ? b2 : best.w2;
}

template <int KIND>
__global__ __launch_bounds__(kBlock, PTSS_MINWAVES) void loopKernel(const float4* __restrict__ blob, SceneLayout L, float* out, int reps) {
    extern __shared__ float4 lds[];
    for (int k = threadIdx.x; k < L.ldsVec4; k += kBlock) lds[k] = blob[k];
    __syncthreads();
    const float4* sc = lds;
    // a ray per lane: origin inside the box, unit direction (a cheap hash; quality is irrelevant here)
    uint32_t h = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (float)(h >> 8) * (1.0f / 16777216.0f); };
    vec3 o = v3(rnd() * 4 - 2, rnd() * 4 - 2, -1 - rnd() * 7);
    vec3 d = normalize(v3(rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f));
    float acc = 0;
    uint32_t iacc = 0;
    for (int r = 0; r < reps; ++r) {
        if (KIND == 1) {  // sphere candidate mask of every sphere
            for (int base = 0; base < L.numSpheres; base += 32) {
                const int cnt = (L.numSpheres - base < 32) ? (L.numSpheres - base) : 32;
#if PTSS_SPHERE_UNROLL & 1
                uint32_t mask = sphereCandidates<false>(sc + L.offSphere + base, cnt, o, d) & lowBits(cnt);
#else
                uint32_t mask = 0;
                for (int j = 0; j < cnt; ++j)
                    if (sphereMayHit(sc[L.offSphere + base + j], o, d)) mask |= 1u << j;
#endif
                iacc += mask;
            }
        }
        if (KIND == 2) {  // closest hit, generic
            const Hit hit = closestHit<false, false>(sc, blob, L, o, d, true, nullptr);
            acc += hit.distance == ptm::inf() ? 0.0f : hit.distance;
            iacc += (uint32_t)hit.idx;
        }
        if (KIND == 3) {  // any hit, dense
            const bool occ = anyHit<false>(sc, L, o, d, 3.0f, true);
            iacc += occ ? 1u : 0u;
        }
        if (KIND == 4) {  // triangles of the closest hit only
            const unsigned long long liveMask = maskOf(true);
            float limit = ptm::inf();
            for (int i = 0; i < L.numTriangles; ++i) {
                const TriRows tcur = loadTri(sc + L.offTri + 3 * i);
                const TriHit th = triangleTest(tcur, o, d, limit, liveMask);
                if (th.hit) { limit = th.dist; iacc = (uint32_t)i; acc = th.w0 + th.w1; }
            }
            acc += limit == ptm::inf() ? 0.0f : limit;
        }
        if (KIND == 6 || KIND == 7) {  // straight-line triangle loop, with (6) / without (7) the reciprocal's range guard
            Best best{ptm::inf(), 0, 0, 0, 0};
            for (int i = 0; i < L.numTriangles; ++i) {
                const TriRows tcur = loadTri(sc + L.offTri + 3 * i);
                triStraight<KIND == 6>(tcur, i, o, d, best);
            }
            acc += (best.dist == ptm::inf() ? 0.0f : best.dist) + best.w0 + best.w1;
            iacc += (uint32_t)best.idx;
        }
        if (KIND == 8 || KIND == 9) {  // the same, two triangles per trip (9: four)
            Best best{ptm::inf(), 0, 0, 0, 0};
            constexpr int U = KIND == 8 ? 2 : 4;
            for (int i = 0; i + U <= L.numTriangles; i += U) {
                TriRows t[U];
#pragma unroll
                for (int k = 0; k < U; ++k) t[k] = loadTri(sc + L.offTri + 3 * (i + k));
#pragma unroll
                for (int k = 0; k < U; ++k) triStraight<false>(t[k], i + k, o, d, best);
            }
            acc += (best.dist == ptm::inf() ? 0.0f : best.dist) + best.w0 + best.w1;
            iacc += (uint32_t)best.idx;
        }
        if (KIND == 10) {  // VALU work of the straight-line loop without its LDS reads (rows invented from the index)
            Best best{ptm::inf(), 0, 0, 0, 0};
            TriRows t = loadTri(sc + L.offTri);
            for (int i = 0; i < L.numTriangles; ++i) {
                t.a.x += 1.0f;  // three instructions: keep the iterations distinct and nothing loop-invariant
                t.b.y += 0.5f;
                t.c.z -= 0.25f;
                triStraight<false>(t, i, o, d, best);
            }
            acc += (best.dist == ptm::inf() ? 0.0f : best.dist) + best.w0 + best.w1;
            iacc += (uint32_t)best.idx;
        }
        if (KIND == 11) {  // LDS reads of the loop alone (three rows per triangle, summed)
            float sum = 0;
            for (int i = 0; i < L.numTriangles; ++i) {
                const TriRows t = loadTri(sc + L.offTri + 3 * i);
                sum += t.a.x + t.b.y + t.c.z;
            }
            acc += sum;
        }
        if (KIND == 12 || KIND == 13 || KIND == 14 || KIND == 15) {  // 64 independent plain VALU instructions per call
            float a[8] = {o.x, o.y, o.z, d.x, d.y, d.z, acc, 1.0f};
            uint32_t u[8] = {h, h + 1, h + 2, h + 3, iacc, iacc + 5, 7u, 9u};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (KIND == 12) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[c]) : "v"(d.x), "v"(d.y));
                    if (KIND == 13) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(u[c]) : "v"(h));
                    if (KIND == 14) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(a[c]) : "v"(d.x));
                    if (KIND == 15) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[c]) : "v"(d.x), "v"(d.y));
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) { acc += a[c]; iacc += u[c]; }
        }
        if (KIND == 5) {  // XORWOW draws: 8 per call
            ptrng::State s;
            s.v[0] = h; s.v[1] = h ^ 0x1234567u; s.v[2] = iacc; s.v[3] = 77u; s.v[4] = h * 3u; s.d = 5u;
            for (int k = 0; k < 8; ++k) acc += ptrng::uniform(s);
            iacc += s.v[4];
        }
        // perturb the ray so that nothing can be hoisted out of the repetition loop
        o.x += 1e-3f;
        d = v3(d.y, d.z, d.x);
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc + (float)iacc;
}

template <int KIND>
static void run(const char* name, const float4* dBlob, const SceneLayout& L, float* dOut, int blocks, int reps, int prims) {
    const size_t lds = bounceLdsBytes(L, true);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(loopKernel<KIND>, dim3(blocks), dim3(kBlock), lds, 0, dBlob, L, dOut, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(loopKernel<KIND>, dim3(blocks), dim3(kBlock), lds, 0, dBlob, L, dOut, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double waveCalls = (double)blocks * kWaves * reps;
    const double simdCycles = ms * 1e-3 * 2.4e9 * 1024.0;  // 1,024 SIMDs at 2.4 GHz
    printf("%-28s %8.3f ms  %8.1f SIMD-cycles per wave-call  (%6.2f per primitive)\n", name, ms, simdCycles / waveCalls,
           simdCycles / waveCalls / prims);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 22, T = argc > 2 ? atoi(argv[2]) : 16;
    SceneLayout L{};
    L.numSpheres = S;
    L.numTriangles = T;
    int off = 0;
    L.offSphere = off; off += (S + 3) / 4 * 4;
    L.offSphereMat = off; off += (S + 3) / 4;
    L.offTri = off; off += 3 * T;
    L.offTriNormal = off; off += 3 * T;
    L.offTriVert = off; off += 2 * T;
    L.offMaterial = off; off += 5 * 14;
    L.offAreaLight = off; off += 2;
    L.offQuant = off; off += 65;
    L.offPrimSphere = off; off += (S + 3) / 4 * 4;
    L.offPrimTri = off; off += 2 * T;
    L.ldsVec4 = L.totalVec4 = off;
    std::vector<float4> blob(off + 1, float4{0, 0, 0, 0});
    srand(1);
    auto u = []() { return rand() / (float)RAND_MAX; };
    for (int i = 0; i < S; ++i) {
        const float r = 0.3f + 0.5f * u();
        blob[L.offSphere + i] = float4{u() * 5 - 2.5f, u() * 5 - 2.5f, -2 - u() * 7, r * r};
    }
    // the walls of a box [-3,3] x [-3,3] x [-10,0] as triangle pairs, then random triangles
    const float X = 3, Z0 = 0, Z1 = -10;
    const float quads[6][4][3] = {{{-X, -X, Z0}, {X, -X, Z0}, {-X, -X, Z1}, {X, -X, Z1}},  {{-X, X, Z0}, {X, X, Z0}, {-X, X, Z1}, {X, X, Z1}},
                                  {{-X, -X, Z0}, {-X, X, Z0}, {-X, -X, Z1}, {-X, X, Z1}},  {{X, -X, Z0}, {X, X, Z0}, {X, -X, Z1}, {X, X, Z1}},
                                  {{-X, -X, Z1}, {X, -X, Z1}, {-X, X, Z1}, {X, X, Z1}},    {{-X, -X, Z0}, {X, -X, Z0}, {-X, X, Z0}, {X, X, Z0}}};
    for (int i = 0; i < T; ++i) {
        float v[3][3];
        if (i < 12) {
            const int q = i / 2;
            const int idx[2][3] = {{0, 1, 2}, {3, 1, 2}};
            for (int k = 0; k < 3; ++k)
                for (int c = 0; c < 3; ++c) v[k][c] = quads[q][idx[i & 1][k]][c];
        } else {
            for (int k = 0; k < 3; ++k) { v[k][0] = u() * 4 - 2; v[k][1] = u() * 4 - 2; v[k][2] = -1 - u() * 8; }
        }
        blob[L.offTri + 3 * i] = float4{v[0][0], v[0][1], v[0][2], 0};
        blob[L.offTri + 3 * i + 1] = float4{v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2], 0};
        blob[L.offTri + 3 * i + 2] = float4{v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2], 0};
    }
    float4* dBlob;
    float* dOut;
    const int blocks = 256 * 7 * 4;
    hipMalloc(&dBlob, blob.size() * sizeof(float4));
    hipMemcpy(dBlob, blob.data(), blob.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&dOut, (size_t)blocks * kBlock * sizeof(float));
    int perCU = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, loopKernel<2>, kBlock, bounceLdsBytes(L, true));
    printf("scene %d spheres, %d triangles; %zu B of LDS per workgroup, %d workgroups per CU; SU=%d ROW128=%d\n", S, T,
           bounceLdsBytes(L, true), perCU, PTSS_SPHERE_UNROLL, PTSS_ROW128);
    const int reps = argc > 3 ? atoi(argv[3]) : 40;
    run<1>("sphere candidate masks", dBlob, L, dOut, blocks, reps, S);
    run<4>("triangle loop (closest)", dBlob, L, dOut, blocks, reps, T);
    run<2>("closestHit", dBlob, L, dOut, blocks, reps, S + T);
    run<3>("anyHit (dense)", dBlob, L, dOut, blocks, reps, S + T);
    run<5>("8 XORWOW uniforms", dBlob, L, dOut, blocks, reps, 8);
    run<6>("tri straight-line, guarded rcp", dBlob, L, dOut, blocks, reps, T);
    run<7>("tri straight-line, bare rcp", dBlob, L, dOut, blocks, reps, T);
    run<8>("tri straight, bare, 2 per trip", dBlob, L, dOut, blocks, reps, T);
    run<9>("tri straight, bare, 4 per trip", dBlob, L, dOut, blocks, reps, T);
    run<12>("64 x v_fmac_f32_e32", dBlob, L, dOut, blocks, reps, 64);
    run<13>("64 x v_xor_b32_e32", dBlob, L, dOut, blocks, reps, 64);
    run<14>("64 x v_mul_f32_e64 (neg)", dBlob, L, dOut, blocks, reps, 64);
    run<15>("64 x v_fma_f32 (VOP3, neg)", dBlob, L, dOut, blocks, reps, 64);
    run<10>("tri straight, no LDS reads", dBlob, L, dOut, blocks, reps, T);
    run<11>("tri rows: LDS reads only", dBlob, L, dOut, blocks, reps, T);
    return 0;
}
