// xorwow.h — the per-pixel random stream of the renderer: Marsaglia XORWOW with cuRAND-style
// seeding and 2^67-long subsequences.
//
// Replaces the reference's use of cuRAND (not vendored; CUDA 6.5):
//   curand_init(seed, sequence=offset, 0, &state[offset])   CudaTracer/CudaTracer.cu:28
//   curand_uniform(&state)  x13 call sites                  CudaTracer/CudaTracer.cu:211,327,328,400-402,408,...
// The generator is public (Marsaglia 2003, "Xorshift RNGs", xorwow); the seed scramble constants
// and the (0,1] float mapping follow cuRAND's published header semantics [unverifiable offline,
// pinned here]. The 2^67 subsequence jump is computed, not copied: A^(2^67) is obtained by squaring
// the 160x160 GF(2) transition matrix 67 times at context creation (host), and
// tests/test_xorwow.py checks it against rocRAND's independent precomputed tables.
//
// State layout in HBM is SoA: six uint32 planes (v0..v4, d), bound to the PIXEL (SURVEY.md §9.2
// decision), not to the compacted slot.
#pragma once
#include <stdint.h>
#include "ptmath.h"

namespace ptrng {

constexpr int kBits = 160;
constexpr int kWords = 5;
constexpr int kJumpLevels = 32;                       // subsequence index < 2^32
constexpr int kJumpTableWords = kJumpLevels * kBits * kWords;

struct State {
    uint32_t v[5];
    uint32_t d;
};

PTM_HD uint32_t next(State& s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1];
    s.v[1] = s.v[2];
    s.v[2] = s.v[3];
    s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}

// curand_uniform: (0, 1].  x * 2^-32 + 2^-33, two roundings (no fma).
PTM_HD float uniform(State& s) {
    uint32_t x = next(s);
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

// seed scramble of curand_init; subsequence 0, offset 0.
PTM_HD State seeded(uint64_t seed) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    State st;
    st.d = 6615241u + t1 + t0;
    st.v[0] = 123456789u + t0;
    st.v[1] = 362436069u ^ t0;
    st.v[2] = 521288629u + t1;
    st.v[3] = 88675123u ^ t1;
    st.v[4] = 5783321u + t0;
    return st;
}

// v <- M v over GF(2); M is stored as the images of the 160 unit vectors (5 words each).
PTM_HD void apply(const uint32_t* __restrict__ m, uint32_t v[5]) {
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    for (int w = 0; w < kWords; ++w) {
        uint32_t bits = v[w];
        for (int b = 0; b < 32; ++b) {
            uint32_t mask = 0u - ((bits >> b) & 1u);
            const uint32_t* row = m + (w * 32 + b) * kWords;
            r0 ^= row[0] & mask;
            r1 ^= row[1] & mask;
            r2 ^= row[2] & mask;
            r3 ^= row[3] & mask;
            r4 ^= row[4] & mask;
        }
    }
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4;
}

// Skip `subsequence` * 2^67 draws. d is unchanged (362437 * 2^67 = 0 mod 2^32).
PTM_HD void skip_subsequences(State& s, uint32_t subsequence, const uint32_t* __restrict__ table) {
    for (int k = 0; k < kJumpLevels; ++k) {
        if ((subsequence >> k) & 1u) apply(table + k * kBits * kWords, s.v);
    }
}

// Host: table[k] = A^(2^(67+k)), k = 0..31, A = one xorshift step on v[0..4].
inline void build_subsequence_table(uint32_t* table /* kJumpTableWords */) {
    static_assert(sizeof(uint32_t) == 4, "");
    uint32_t* a = new uint32_t[kBits * kWords];
    uint32_t* b = new uint32_t[kBits * kWords];
    for (int i = 0; i < kBits; ++i) {
        State e{};
        e.v[i / 32] = 1u << (i % 32);
        (void)next(e);
        for (int w = 0; w < kWords; ++w) a[i * kWords + w] = e.v[w];
    }
    auto square = [&](const uint32_t* src, uint32_t* dst) {
        for (int i = 0; i < kBits; ++i) {
            uint32_t col[5];
            for (int w = 0; w < kWords; ++w) col[w] = src[i * kWords + w];
            apply(src, col);
            for (int w = 0; w < kWords; ++w) dst[i * kWords + w] = col[w];
        }
    };
    for (int s = 0; s < 67; ++s) {
        square(a, b);
        uint32_t* t = a; a = b; b = t;
    }
    for (int k = 0; k < kJumpLevels; ++k) {
        for (int i = 0; i < kBits * kWords; ++i) table[k * kBits * kWords + i] = a[i];
        square(a, b);
        uint32_t* t = a; a = b; b = t;
    }
    delete[] a;
    delete[] b;
}

}  // namespace ptrng
