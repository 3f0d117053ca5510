// ptss_diag.h — diagnostic counters of the bounce kernel. They exist only in -DPTSS_DIAG=<bits> builds
// (tools/build_variants.py: chist, shist, cullstat, pairstat, qhist); with PTSS_DIAG == 0 every hook below is an empty
// statement and the shipped kernel carries no counter. Included by ptss_kernels.hip inside its anonymous namespace, after
// the chunk-bound helpers it uses. The host reads the eight words with ptss_debug_counters().
#pragma once

#if PTSS_DIAG
__device__ unsigned long long g_diag[8];
#endif

// bit 0 — sphere candidates per lane, wave maximum vs wave mean (tools/candidate_hist.py). slot 0: closest hit, 4: dense any-hit
#if PTSS_DIAG & 1
__device__ __forceinline__ void diagCandidates(uint32_t mask, bool live, int slot) {
    const uint32_t pc = (uint32_t)__builtin_popcount(mask);
    uint32_t mx = 0;
    while (__any(pc > mx)) ++mx;
    uint32_t total = 0;
    for (uint32_t b = 0; b < 6; ++b) total += (uint32_t)__popcll(__ballot((pc >> b) & 1u)) << b;
    if (__lane_id() == 0) {
        atomicAdd(&g_diag[slot + 0], (unsigned long long)mx);
        atomicAdd(&g_diag[slot + 1], (unsigned long long)total);
        atomicAdd(&g_diag[slot + 2], 1ull);
        atomicAdd(&g_diag[slot + 3], (unsigned long long)__popcll(__ballot(live)));
    }
}
#define PTSS_DIAG_CANDIDATES(mask, live, slot) diagCandidates(mask, live, slot)
#else
#define PTSS_DIAG_CANDIDATES(mask, live, slot) do {} while (0)
#endif

// bit 1 — how many waves execute each block of scatter(), and for how many lanes (tools/scatter_hist.py)
#if PTSS_DIAG & 2
#define PTSS_DIAG_SCATTER(k, cond)                                                             \
    do {                                                                                       \
        const unsigned long long _m = __ballot(cond);                                          \
        if (_m != 0ull && __lane_id() == (unsigned)(__ffsll((long long)__ballot(true)) - 1)) { \
            atomicAdd(&g_diag[k], 1ull + ((unsigned long long)__popcll(_m) << 32));            \
        }                                                                                      \
    } while (0)
#else
#define PTSS_DIAG_SCATTER(k, cond) do {} while (0)
#endif

// bit 2 — many-sphere scenes: of the chunks a ray's line touches, how many are NOT wholly beyond its final hit (tools/cull_stat.py)
#if PTSS_DIAG & 4
__device__ __forceinline__ void diagCull(const float4* sc, const SceneLayout& L, vec3 o, vec3 d, bool unitDir, bool live, unsigned long long won) {
    const float T = (won == ~0ull) ? ptm::inf() : asF((uint32_t)(won >> 32));
    uint32_t touched = 0, needed = 0;
    for (int g = 0; g < L.numChunks && g < 128; g += 32) {
        const int left = L.numChunks - g;
        const uint32_t bits = live ? chunkMask<false>(sc + L.offChunk + g, left < 32 ? left : 32, o, d, unitDir) : 0u;
        uint32_t rev = 0;
        const int trips = ((left < 32 ? left : 32) + 3) >> 2;
        for (int q = 0; q < 4 * trips; ++q) {
            const float4 b = sc[L.offChunk + g + q];
            const vec3 v = o - xyz(b);
            const float dv = dot(d, v), vv = dot(v, v), a = -dv - T;
            const bool ahead = unitDir && (a > 0.0f) && ((a * a) * (1.0f - 2e-5f) > (1.0f - kAccelMu) * b.w + kAccelMu * vv);
            rev |= ahead ? 0u : (1u << q);
        }
        touched += (uint32_t)__builtin_popcount(bits);
        needed += (uint32_t)__builtin_popcount(bits & rev);
    }
    uint32_t st = 0, sn = 0;
    for (uint32_t bit = 0; bit < 7; ++bit) {
        st += (uint32_t)__popcll(__ballot((touched >> bit) & 1u)) << bit;
        sn += (uint32_t)__popcll(__ballot((needed >> bit) & 1u)) << bit;
    }
    const uint32_t hits = (uint32_t)__popcll(__ballot(live && won != ~0ull));
    const uint32_t rays = (uint32_t)__popcll(__ballot(live));
    if (__lane_id() == 0) {
        atomicAdd(&g_diag[0], (unsigned long long)st);
        atomicAdd(&g_diag[1], (unsigned long long)sn);
        atomicAdd(&g_diag[2], (unsigned long long)rays);
        atomicAdd(&g_diag[3], (unsigned long long)hits);
    }
}
#define PTSS_DIAG_CULL(sc, L, o, d, unitDir, live, won) diagCull(sc, L, o, d, unitDir, live, won)
#else
#define PTSS_DIAG_CULL(sc, L, o, d, unitDir, live, won) do {} while (0)
#endif

// bit 3 — lit lanes per wave and NEE round that queue both / one of their two shadow segments (tools/pair_stat.py)
#if PTSS_DIAG & 8
#define PTSS_DIAG_PAIRS(needA, needB, lit)                                                                              \
    do {                                                                                                                \
        const unsigned long long _mb = __ballot((needA) && (needB)), _mo = __ballot((needA) != (needB)), _ml = __ballot(lit); \
        if (__lane_id() == 0) {                                                                                         \
            atomicAdd(&g_diag[4], (unsigned long long)__popcll(_mb));                                                   \
            atomicAdd(&g_diag[5], (unsigned long long)__popcll(_mo));                                                   \
            atomicAdd(&g_diag[6], (unsigned long long)__popcll(_ml));                                                   \
            atomicAdd(&g_diag[7], 1ull);                                                                                \
        }                                                                                                               \
    } while (0)
#else
#define PTSS_DIAG_PAIRS(needA, needB, lit) do {} while (0)
#endif

// bit 4 — histogram of the wave's shadow-queue length per NEE round (tools/queue_hist.py)
#if PTSS_DIAG & 16
#define PTSS_DIAG_QUEUE(queued)                                                                                               \
    do {                                                                                                                      \
        if (__lane_id() == 0)                                                                                                 \
            atomicAdd(&g_diag[(queued) == 0 ? 0 : ((queued) <= 8 ? 1 : ((queued) <= 16 ? 2 : ((queued) <= 32 ? 3 : ((queued) <= 64 ? 4 : ((queued) <= 72 ? 5 : ((queued) <= 96 ? 6 : 7))))))], 1ull); \
    } while (0)
#else
#define PTSS_DIAG_QUEUE(queued) do {} while (0)
#endif
