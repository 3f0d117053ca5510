// ptss_api.hip — context management and the frame driver behind include/ptss.h.
//
// ptss_generate_frame is the reference's generateFrame (CudaTracer/CudaTracer.cu:587-647) with the
// host taken out of the inner loop: the per-bounce live-ray count stays on the device
// (FrameBuffers::counts), every bounce kernel is launched unconditionally and applies the
// reference's `numRays > 128` guard itself, so a frame is one uninterrupted stream of launches
// with at most one event wait at the end (cfg.syncEachFrame, the reference's :639-642).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "ptss.h"
#include "ptss_device.h"
#include <algorithm>
#include <cmath>
#include "ptquant.h"
#include "pttri.h"

using namespace ptv;

#if PTSS_DIAG
namespace ptss { hipError_t readDiagCounters(unsigned long long* out8); }
#endif
namespace {

thread_local std::string g_detail;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    g_detail = buf;
    return code;
}

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return fail(PTSS_EHIP, #expr, _e); \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

// One ray population of the frame with its own pools, counters and stream (ptss_device.h, "frame lanes"). A context
// has 1..kMaxLanes of them; with one lane the caller's stream is used and nothing below differs from a single population.
struct Lane {
    hipStream_t stream = nullptr;            // own stream (contexts with several lanes only)
    float* dPool[2] = {nullptr, nullptr};
    uint32_t* dCounts[2] = {nullptr, nullptr};   // alternate per frame (flushKernel arms the other one)
    uint32_t* dShardCount0 = nullptr;
    uint32_t* dLastCounts = nullptr;         // counts of the frame before (flushKernel's copy)
    uint32_t* dDone = nullptr;               // done[countIndex(b, s)]: workgroups of shard s that ended a bounce-b kernel, all frames (never reset)
    uint32_t doneTarget[ptss::kMaxBounces + 1] = {};  // what done[b][*] add up to once everything launched so far has ended
                                             // (stored by this lane's bounce-b kernel as it starts: bounce b - 1 has then finished)
    hipEvent_t evDone[2] = {nullptr, nullptr};   // end of this lane's frame, by frame parity (several lanes only)
    uint32_t regionCap = 0;                  // slots per shard region of this lane's pools
    int maxBlocks = 0;                       // one 256-ray tile per workgroup over this lane's share of the frame
    // live-count hints: counts[] of a recent frame, read back asynchronously, size the next frames' grids
    uint32_t hint[ptss::kMaxBounces + 1] = {0};  // per bounce: the fullest shard's live count
    bool haveHint = false;
    uint32_t* hCounts = nullptr;  // pinned, 4 slots x kCountWords
    hipEvent_t hintEvent[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hintPending[4] = {false, false, false, false};
};

struct ptss_context {
    ptss_render_config cfg{};
    hipStream_t stream = nullptr;
    ptss::TileMap tile{};
    ptss::SceneLayout layout{};
    float4* dScene = nullptr;       // the scene image the frames use
    float4* dSceneAlt = nullptr;    // with sphere acceleration on: the plain image, for cameras outside its range
    ptss::SceneLayout layoutAlt{};
    bool sceneInLdsAlt = true;
    bool haveAccel = false, accelActive = false;
    std::vector<Lane> lanes;
    hipEvent_t evFork = nullptr;            // several lanes: the caller's stream has reached this frame
    int countParity = 0;                    // which of a lane's two count buffers the next frame uses
    uint32_t* dRngHome = nullptr;
    unsigned long long* dTotal = nullptr;   // [0 .. kMaxLanes) ray-bounce totals per lane, [kMaxLanes .. +8) unused,
                                            // then one word: guard timeouts (must stay 0)
    uint32_t* dAccumOwned = nullptr;
    uint32_t* dAccum = nullptr;  // owned or bound
    float* dFsum = nullptr;
    uint32_t* dStaged = nullptr;  // S > 1: per-stream sample words of the current pass. Free-running lanes: TWO buffers, by frame parity —
                                  // the lanes of frame N + 1 already park samples while displayKernel of frame N (on the caller's stream,
                                  // behind the join) still adds up frame N's; a lane starts frame N + 2 only behind displayKernel of frame N
    bool stagedTwice = false;
    hipEvent_t evDisplay[2] = {nullptr, nullptr};   // displayKernel of the last frame of each parity has run (stagedTwice only)
    uint32_t capacity = 0, numPixels = 0;  // capacity: stride of the per-pixel planes (rngHome)
    uint32_t samples = 1;                    // cfg.samplesPerPass (sample lanes per pixel)
    bool cameraDirty = true;           // primary-ray precomputes must be refreshed
    float defaultColor[3] = {0, 0, 0};
    // ProgramData (CudaTracer.h:32-42)
    ptss_camera camera{};
    int lastResetTick = 0;
    int lastTicks = 0;
    unsigned maxIterations = 15;
    bool resetTicksThisFrame = true;
    bool usePathTracer = true;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    float lastMs = 0.0f;
    bool oneLaunch = false, oneLaunchAlt = false;   // the frame is traced by ONE launch (frameKernel) with the current / the alternate scene image
    int gridCap = 0;             // workgroups per shard at most = 16 resident rounds of this scene's bounce kernel (0 = uncapped)
    bool sceneInLds = true;      // scene staged in LDS (true) or read through scalar loads (false)
    unsigned frameIndex = 0;
    // bounce-kernel timing (cfg.timeKernels): every launch is bracketed by two events on ITS stream; a finished pair becomes
    // an interval [start, end) in ms since evEpoch. ptss_bounce_kernel_time reports the UNION of the intervals: with one lane
    // that is the sum of the launch durations, with several lanes (whose kernels overlap in time) the time during which at
    // least one bounce kernel was running — the figure a launch-time roofline needs.
    std::vector<EventPair> evFree, evBusy;
    std::vector<std::pair<double, double>> kernelSpans;
    hipEvent_t evEpoch = nullptr;
    unsigned int timeoutsSeen = 0;   // ptss_guard_timeouts value already reported as PTSS_ETIMEOUT
};
constexpr int kTotalWords = ptss::kMaxLanes + 8 + 1;

namespace {

// ---- sphere acceleration (scenes with many spheres) ------------------------------------------------------------------
// The kernel may skip a sphere only if the reference's test certainly rejects it (Primitives.h:107-127, float32): its
// discriminant is certainly negative, or both its roots are. Spheres are sorted spatially (spatialOrder) and cut into chunks of kChunkSpheres; each chunk gets
// a bounding sphere (C, R) with |c_i - C| + r_i <= R for its members. For a ray (o, d) with | |d|^2 - 1 | <= eps = 1e-5
// let vC = o - C, vv = vC.vC, dv = d.vC, dm = min(dv, 0). The kernel culls the chunk iff
//       vv - (1 + 2 eps) / (1 - mu) dm^2  >  R^2 (1 + m)^3 (1 + 4e-6) / (1 - mu)        with m = 5e-3, mu = m + m^2
// (shiftInChunk: the right side is the stored bound, the factor of dm^2 is 4 kAccelQ, both rounded up).
// Why that is safe. Let E be the distance of C from the RAY {o + t d^, t >= 0}: E^2 = vv - dm^2 / |d|^2 (the line's distance
// while the closest approach lies ahead, |vC| once it lies behind the origin), and 1 / |d|^2 <= 1 + 2 eps. Multiplied by
// (1 - mu) the test says E^2 - mu vv > R^2 (1+m)^3 (1 + 4e-6) in real arithmetic; the float evaluation of the left side
// (v rounded per component, two three-term dot products, t = dv - |dv| exact, one product, one fma) errs by less than
// 1e-6 vv, which the factor (1 + 4e-6) pays for even where the next step has no slack (|vC| = R (1 + m)):
// R^2 (1+m)^3 + mu vv >= (R + m (|vC| + R))^2 (AM-GM), so E > R + m (|vC| + R). The distance of a member's centre from the
// ray is then E_i >= E - |c_i - C| > r_i + m |v_i| (|v_i| <= |vC| + R), i.e. E_i^2 > r_i^2 + 2.5e-5 |v_i|^2. Two cases.
// The member's closest approach lies ahead (d.v_i <= 0): E_i is the line's distance dist_i, and the exact discriminant / 4,
// r_i^2 - dist_i^2 + (|d|^2 - 1)(d^.v_i)^2 <= r_i^2 - dist_i^2 + 1e-5 |v_i|^2, is below -1.5e-5 |v_i|^2; float32 evaluation
// moves it by less than 1e-6 |v_i|^2 (|v_i| > r_i here): negative, Sphere::intersectRay returns false (Primitives.h:118).
// It lies behind (d.v_i > 0): E_i = |v_i|, so c = |v_i|^2 - r_i^2 > 2.5e-5 |v_i|^2, in float32 still > 2.4e-5 |v_i|^2. If
// the float discriminant b^2 - 4c is negative the test returns false; if not, b^2 >= 4c > 9.6e-5 |v_i|^2 puts |b| far above
// its rounding error (5e-7 |v_i|), so b has its true sign, positive, and 4c >= 2.4e-5 b^2 keeps sqrt(b^2 - 4c) below
// b (1 - 1e-5): both roots (-b +- sqrt) / 2 are negative beyond any rounding and the test returns false (Primitives.h:123-127).
// Rays whose direction is not unit to 1e-5 (the reference does not renormalise blended vertex normals) visit every chunk;
// a NaN anywhere fails the `>`. Until round 3 the kernel tested the LINE's distance and, separately, "the bound lies wholly
// behind the plane through the origin": the ray's distance is one test instead of two and skips more — a ray that leaves
// a chunk it starts beside no longer enters it (tools/chunk_bounds_stat.py: 5.09 -> 4.67 chunks per mid-bounce ray).
// Requires finite, moderate geometry (|coordinate|, radius <= 1e15, radius >= 1e-12) so that no discriminant overflows; packScene and
// the per-frame camera check fall back to the plain image otherwise.
constexpr double kAccelM = 5e-3;
constexpr float kAccelLimit = 1e15f;
constexpr int kAccelMinSpheres = 64;

// Finite, moderate geometry: every |coordinate| <= 1e15, every sphere radius in [1e-12, 1e15] (false for NaN and infinities).
// What the chunked traversal requires, and what lets the sphere candidate tests take their shorter form
// (SceneLayout::sphereBounded, ptss_kernels.hip shiftInSphere<true>: r^2 well inside the normal range, no discriminant near overflow).
bool geometryBounded(const ptss_scene_desc& s) {
    auto ok = [](float v) { return std::fabs(v) <= kAccelLimit; };
    for (size_t i = 0; i < s.numSpheres; ++i) {
        const ptss_sphere& sp = s.spheres[i];
        if (!ok(sp.position.x) || !ok(sp.position.y) || !ok(sp.position.z) || !ok(sp.radius)) return false;
        if (!(std::fabs(sp.radius) >= 1e-12f)) return false;
    }
    for (size_t i = 0; i < s.numTriangles; ++i) {
        const ptss_triangle& t = s.triangles[i];
        for (const ptss_vec3* v : {&t.vertex0, &t.vertex1, &t.vertex2})
            if (!ok(v->x) || !ok(v->y) || !ok(v->z)) return false;
    }
    for (size_t i = 0; i < s.numPointLights; ++i)
        if (!ok(s.pointLights[i].position.x) || !ok(s.pointLights[i].position.y) || !ok(s.pointLights[i].position.z)) return false;
    return true;
}
bool cameraInRange(const ptss_camera& cam) {
    auto ok = [](float v) { return std::fabs(v) <= kAccelLimit; };
    return ok(cam.position.x) && ok(cam.position.y) && ok(cam.position.z);
}

bool accelEligible(const ptss_scene_desc& s) { return s.numSpheres >= (size_t)kAccelMinSpheres && geometryBounded(s); }

// sorted position -> original index. The spheres are split recursively at the median of their centres along the axis of
// largest extent (a kd-tree built by std::nth_element; ties by original index, so the order is deterministic), the cut
// placed at a multiple of 64 spheres while a part holds more than 64 and at a multiple of kChunkSpheres below that: every
// chunk of kChunkSpheres consecutive positions is a leaf and every 64 consecutive positions a subtree. Against round 1's
// Morton curve (whose jumps put far-apart spheres into one chunk) a ray of the configs[5] scene meets about half as many
// chunk bounds. Any permutation is legal here: the traversal decides ties by ORIGINAL index (offSphereOrig).
void kdSplit(const ptss_scene_desc& s, std::vector<int>& idx, int lo, int hi) {
    const int n = hi - lo;
    if (n <= ptss::kChunkSpheres) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i) {
        const float c[3] = {s.spheres[idx[i]].position.x, s.spheres[idx[i]].position.y, s.spheres[idx[i]].position.z};
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], c[a]);
            mx[a] = std::max(mx[a], c[a]);
        }
    }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    const int unit = n > 64 ? 64 : ptss::kChunkSpheres;
    int left = ((n / 2 + unit - 1) / unit) * unit;  // spheres in the lower part: about half, a whole number of units
    if (left >= n) left -= unit;
    if (left <= 0) return;
    auto key = [&](int i) { return axis == 0 ? s.spheres[i].position.x : (axis == 1 ? s.spheres[i].position.y : s.spheres[i].position.z); };
    std::nth_element(idx.begin() + lo, idx.begin() + lo + left, idx.begin() + hi,
                     [&](int a, int b) { return key(a) < key(b) || (key(a) == key(b) && a < b); });
    kdSplit(s, idx, lo, lo + left);
    kdSplit(s, idx, lo + left, hi);
}
// A ball around n member spheres: from the mean of their centres, `steps` steps of "move towards the farthest point of the
// farthest member by 1 / (step + 1) of the way" (Badoiu-Clarkson), keeping the best centre seen — close to the smallest
// enclosing ball. Any centre is legal for a chunk bound: packScene measures R from the float centre it stores.
struct Ball {
    double c[3], r;
};
Ball enclosingBall(const ptss_scene_desc& s, const int* idx, int n, int steps) {
    auto reach = [&](const double c[3], int j, double* toward) {   // distance from c to the far side of member j
        const ptss_sphere& sp = s.spheres[idx[j]];
        const double dx = (double)sp.position.x - c[0], dy = (double)sp.position.y - c[1], dz = (double)sp.position.z - c[2];
        const double len = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (toward) { toward[0] = len > 0 ? dx / len : 0; toward[1] = len > 0 ? dy / len : 0; toward[2] = len > 0 ? dz / len : 0; }
        return len + std::fabs((double)sp.radius);
    };
    double C[3] = {0, 0, 0};
    for (int j = 0; j < n; ++j) {
        C[0] += s.spheres[idx[j]].position.x / n; C[1] += s.spheres[idx[j]].position.y / n; C[2] += s.spheres[idx[j]].position.z / n;
    }
    Ball best{{C[0], C[1], C[2]}, INFINITY};
    for (int step = 1; step <= steps; ++step) {
        int far = 0;
        double farR = -1, dir[3];
        for (int j = 0; j < n; ++j) {
            const double r = reach(C, j, nullptr);
            if (r > farR) { farR = r; far = j; }
        }
        if (!(farR < INFINITY)) break;
        if (farR < best.r) best = Ball{{C[0], C[1], C[2]}, farR};
        reach(C, far, dir);
        for (int a = 0; a < 3; ++a) C[a] += dir[a] * farR / (step + 1);
    }
    return best;
}
// The kd leaves, improved pair by pair: the members of a chunk and of one of its six nearest chunks are split again, half
// and half, along the line joining the two centres, the axes and three diagonals, and the split with the smallest
// R_a^2 + R_b^2 (the two balls' cross-sections, what a passing line sees) replaces the pair if it beats the present one.
// Up to three sweeps over scenes of up to 256 chunks, one up to 1,024, none beyond (the cost grows with the chunk count and
// is paid at scene set-up). configs[4]'s scene: a mid-bounce ray touches 3.73 bounds instead of 3.95.
void refineChunks(const ptss_scene_desc& s, std::vector<int>& order) {
    constexpr int kM = ptss::kChunkSpheres;
    const int K = (int)(order.size() / kM);   // whole chunks only
    const int sweeps = K < 2 ? 0 : (K <= 256 ? 3 : (K <= 1024 ? 1 : 0));
    if (sweeps == 0) return;
    std::vector<Ball> ball((size_t)K);
    for (int k = 0; k < K; ++k) ball[(size_t)k] = enclosingBall(s, &order[(size_t)k * kM], kM, 64);
    auto centre = [&](int i, int a) { return a == 0 ? (double)s.spheres[i].position.x : (a == 1 ? (double)s.spheres[i].position.y : (double)s.spheres[i].position.z); };
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        int improved = 0;
        for (int a = 0; a < K; ++a) {
            std::vector<std::pair<double, int>> near;
            for (int b = 0; b < K; ++b) {
                if (b == a) continue;
                double d2 = 0;
                for (int x = 0; x < 3; ++x) d2 += (ball[(size_t)b].c[x] - ball[(size_t)a].c[x]) * (ball[(size_t)b].c[x] - ball[(size_t)a].c[x]);
                near.emplace_back(d2, b);
            }
            const size_t take = std::min<size_t>(6, near.size());
            std::partial_sort(near.begin(), near.begin() + take, near.end());
            for (size_t q = 0; q < take; ++q) {
                const int b = near[q].second;
                int both[2 * kM], trial[2 * kM], keep[2 * kM];
                for (int j = 0; j < kM; ++j) { both[j] = order[(size_t)a * kM + j]; both[kM + j] = order[(size_t)b * kM + j]; }
                double bestCost = ball[(size_t)a].r * ball[(size_t)a].r + ball[(size_t)b].r * ball[(size_t)b].r;
                const double floor = bestCost * (1 - 1e-9);
                Ball keepA{}, keepB{};
                bool found = false;
                const double join[3] = {ball[(size_t)b].c[0] - ball[(size_t)a].c[0], ball[(size_t)b].c[1] - ball[(size_t)a].c[1], ball[(size_t)b].c[2] - ball[(size_t)a].c[2]};
                const double dirs[7][3] = {{join[0], join[1], join[2]}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}};
                for (const auto& dir : dirs) {
                    std::copy(both, both + 2 * kM, trial);
                    auto key = [&](int i) { return centre(i, 0) * dir[0] + centre(i, 1) * dir[1] + centre(i, 2) * dir[2]; };
                    std::sort(trial, trial + 2 * kM, [&](int x, int y) { return key(x) < key(y) || (key(x) == key(y) && x < y); });
                    const Ball ta = enclosingBall(s, trial, kM, 48), tb = enclosingBall(s, trial + kM, kM, 48);
                    const double cost = ta.r * ta.r + tb.r * tb.r;
                    if (cost < bestCost && cost < floor) {
                        bestCost = cost; keepA = ta; keepB = tb; found = true;
                        std::copy(trial, trial + 2 * kM, keep);
                    }
                }
                if (found) {
                    for (int j = 0; j < kM; ++j) { order[(size_t)a * kM + j] = keep[j]; order[(size_t)b * kM + j] = keep[kM + j]; }
                    ball[(size_t)a] = keepA; ball[(size_t)b] = keepB;
                    ++improved;
                }
            }
        }
        if (improved == 0) break;
    }
}
std::vector<int> spatialOrder(const ptss_scene_desc& s) {
    std::vector<int> order((size_t)s.numSpheres);
    for (size_t i = 0; i < s.numSpheres; ++i) order[i] = (int)i;
    kdSplit(s, order, 0, (int)s.numSpheres);
    refineChunks(s, order);
    return order;
}

void packScene(const ptss_scene_desc& s, ptss::SceneLayout& L, std::vector<float4>& blob, bool accel) {
    auto u2f = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    L.accelSpheres = accel ? 1 : 0;
    L.numChunks = accel ? (int)((s.numSpheres + ptss::kChunkSpheres - 1) / ptss::kChunkSpheres) : 0;
    const int sphereRows = accel ? L.numChunks * ptss::kChunkSpheres : (int)s.numSpheres;
    L.numSpheres = (int)s.numSpheres;
    L.numTriangles = (int)s.numTriangles;
    L.numMaterials = (int)s.numMaterials;
    L.numPointLights = (int)s.numPointLights;
    L.numAreaLights = (int)s.numAreaLights;
    int off = 0;
    // plain image: sphere rows (and their camera-origin twins) padded to a multiple of four, zero-filled — the
    // candidate pass fetches four rows per trip and drops the padding's bits (sphereCandidates)
    const int sphereAlloc = accel ? sphereRows : (sphereRows + 3) / 4 * 4;
    L.offSphere = off;      off += sphereAlloc;
    if (!accel) { L.offSphereMat = off; off += (sphereRows + 3) / 4; }
    L.offChunk = off;       off += (L.numChunks + 3) / 4 * 4;   // bound rows padded to a multiple of four (zero rows: chunkMask drops their bits)
    L.offTri = off;         off += 3 * L.numTriangles;
    L.offTriNormal = off;   off += 3 * L.numTriangles;
    L.offTriVert = off;     off += 2 * L.numTriangles;
    L.offMaterial = off;    off += 5 * L.numMaterials;
    L.offPointLight = off;  off += 2 * L.numPointLights;
    L.offAreaLight = off;   off += 2 * L.numAreaLights;
    L.offTriPos = off;      off += (L.numTriangles + 3) / 4;   // ints: stored position of each original triangle index
    L.offQuant = off;       off += ptq::kTableFloats / 4;
    L.offPrimSphere = off;  off += accel ? 0 : sphereAlloc;  // the chunked traversal keeps the camera-origin parts of its chunk bounds only (offPrimChunk)
    L.offPrimTri = off;     off += 2 * L.numTriangles;
    L.offPrimChunk = off;   off += accel ? (L.numChunks + 3) / 4 * 4 : 0;
    L.ldsVec4 = off;        // everything up to here is staged into LDS
    if (accel) {            // cold integer tables of the many-sphere image: global memory only
        L.offSphereMat = off;   off += (sphereRows + 3) / 4;
        L.offSphereOrig = off;  off += (sphereRows + 3) / 4;
        L.offSpherePos = off;   off += (L.numSpheres + 3) / 4;
    } else {
        L.offSphereOrig = L.offSpherePos = 0;
    }
    L.totalVec4 = off;
    auto finite3 = [](const ptss_vec3& v) { return v.x - v.x == 0.0f && v.y - v.y == 0.0f && v.z - v.z == 0.0f; };
    L.neeSkipSafe = 1;
    for (size_t i = 0; i < s.numMaterials; ++i)
        if (!finite3(s.materials[i].diffuseColor) || !(s.materials[i].diffAvg - s.materials[i].diffAvg == 0.0f)) L.neeSkipSafe = 0;
    for (size_t i = 0; i < s.numPointLights; ++i)
        if (!finite3(s.pointLights[i].power)) L.neeSkipSafe = 0;
    for (size_t i = 0; i < s.numAreaLights; ++i)
        if (!finite3(s.areaLights[i].power)) L.neeSkipSafe = 0;
    L.sphereBounded = geometryBounded(s) ? 1 : 0;  // see SceneLayout::sphereBounded
    {   // see SceneLayout::neePairs: at least two lights, and at least four of five primitives wear a diffusely reflecting material
        size_t diffuse = 0;
        for (size_t i = 0; i < s.numSpheres; ++i) diffuse += s.materials[s.spheres[i].materialIdx].diffAvg > 0.0f ? 1 : 0;
        for (size_t i = 0; i < s.numTriangles; ++i) diffuse += s.materials[s.triangles[i].materialIdx].diffAvg > 0.0f ? 1 : 0;
        L.neePairs = (L.sphereBounded && s.numPointLights + s.numAreaLights >= 2 && 5 * diffuse >= 4 * (s.numSpheres + s.numTriangles)) ? 1 : 0;
    }
    L.triDetBounded = 1;  // see SceneLayout::triDetBounded
    for (size_t i = 0; i < s.numTriangles; ++i) {
        const ptss_triangle& t = s.triangles[i];
        const vec3 e1 = t.vertex1 - t.vertex0, e2 = t.vertex2 - t.vertex0;  // as stored below
        const double n1 = std::sqrt((double)e1.x * e1.x + (double)e1.y * e1.y + (double)e1.z * e1.z);
        const double n2 = std::sqrt((double)e2.x * e2.x + (double)e2.y * e2.y + (double)e2.z * e2.z);
        if (!(n1 * n2 <= 0x1p100)) L.triDetBounded = 0;  // false for NaN / infinite edges as well
    }
    // Storage order of the triangles: the caller's, or — SceneLayout::triClassed — grouped by edge class (pttri.h), the caller's
    // order kept inside a group. triOrder[position] = original index.
    L.triClassed = (L.triDetBounded && L.sphereBounded && L.numTriangles <= 255) ? 1 : 0;   // (the class bounds travel as bytes)
    std::vector<int> triOrder((size_t)L.numTriangles), triCode((size_t)L.numTriangles, 0);
    for (int i = 0; i < L.numTriangles; ++i) {
        triOrder[(size_t)i] = i;
        const ptss_triangle& t = s.triangles[i];
        if (L.triClassed) triCode[(size_t)i] = pttri::triangleClass(t.vertex1 - t.vertex0, t.vertex2 - t.vertex0);   // the edges as stored below
    }
    std::stable_sort(triOrder.begin(), triOrder.end(), [&](int a, int b) { return triCode[(size_t)a] < triCode[(size_t)b]; });
    for (int k = 0; k < 5; ++k) L.triClassPack[k] = 0u;
    for (int code = 0, pos = 0; code <= 16 && L.triClassed; ++code) {
        while (pos < L.numTriangles && triCode[(size_t)triOrder[(size_t)pos]] < code) ++pos;
        L.triClassPack[code / 4] |= (uint32_t)pos << (8 * (code % 4));
    }
    blob.assign((size_t)off + 1, float4{0, 0, 0, 0});
    ptq::build_thresholds(reinterpret_cast<float*>(&blob[L.offQuant]));
    std::vector<int> order;
    if (accel) {
        order = spatialOrder(s);
        while ((int)order.size() < sphereRows) order.push_back(order.back());  // pad the last chunk with copies
    } else {
        order.resize(s.numSpheres);
        for (size_t i = 0; i < s.numSpheres; ++i) order[i] = (int)i;
    }
    for (int i = 0; i < sphereRows; ++i) {
        const ptss_sphere& sp = s.spheres[order[i]];
        // radius*radius is the same single rounding the reference performs per test (Primitives.h:113)
        blob[L.offSphere + i] = float4{sp.position.x, sp.position.y, sp.position.z, sp.radius * sp.radius};
        reinterpret_cast<int*>(&blob[L.offSphereMat])[i] = sp.materialIdx;
        if (accel) reinterpret_cast<int*>(&blob[L.offSphereOrig])[i] = order[i];
        if (accel && i < L.numSpheres) reinterpret_cast<int*>(&blob[L.offSpherePos])[order[i]] = i;
    }
    for (int k = 0; k < L.numChunks; ++k) {  // bounding sphere of the chunk, in double, rounded outwards
        // Centre: close to the smallest enclosing ball's (enclosingBall) — on the configs[4] scene the radii shrink by 8 % on
        // average (up to 18 %) against the mean of the members' centres, and a mid-bounce ray touches 3.95 instead of 4.67 bounds.
        const Ball ball = enclosingBall(s, &order[(size_t)k * ptss::kChunkSpheres], ptss::kChunkSpheres, 512);
        const float Cf[3] = {(float)ball.c[0], (float)ball.c[1], (float)ball.c[2]};
        double Rmax = 0;
        for (int j = 0; j < ptss::kChunkSpheres; ++j) {
            const ptss_sphere& sp = s.spheres[order[k * ptss::kChunkSpheres + j]];
            const double dx = (double)sp.position.x - Cf[0], dy = (double)sp.position.y - Cf[1], dz = (double)sp.position.z - Cf[2];
            Rmax = std::max(Rmax, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)sp.radius));
        }
        const double infl = Rmax * Rmax * (1 + kAccelM) * (1 + kAccelM) * (1 + kAccelM) * (1 + 4e-6) / (1 - (kAccelM + kAccelM * kAccelM)) * (1 + 1e-9);
        blob[L.offChunk + k] = float4{Cf[0], Cf[1], Cf[2], std::nextafter((float)infl, INFINITY)};
    }
    for (int pos = 0; pos < L.numTriangles; ++pos) {
        const int i = triOrder[(size_t)pos];
        const ptss_triangle& t = s.triangles[i];
        const vec3 e1 = t.vertex1 - t.vertex0;  // Primitives.h:34-35, hoisted (same subtraction, same bits)
        const vec3 e2 = t.vertex2 - t.vertex0;
        blob[L.offTri + 3 * pos + 0] = float4{t.vertex0.x, t.vertex0.y, t.vertex0.z, u2f((uint32_t)t.materialIdx)};
        blob[L.offTri + 3 * pos + 1] = float4{e1.x, e1.y, e1.z, u2f(0xfffffffeu - (uint32_t)i)};   // the low half of the closest hit's (distance, 0xFFFFFFFE - original index) key
        blob[L.offTri + 3 * pos + 2] = float4{e2.x, e2.y, e2.z, 0};
        blob[L.offTriNormal + 3 * pos + 0] = float4{t.normal0.x, t.normal0.y, t.normal0.z, 0};
        blob[L.offTriNormal + 3 * pos + 1] = float4{t.normal1.x, t.normal1.y, t.normal1.z, 0};
        blob[L.offTriNormal + 3 * pos + 2] = float4{t.normal2.x, t.normal2.y, t.normal2.z, 0};
        blob[L.offTriVert + 2 * pos + 0] = float4{t.vertex1.x, t.vertex1.y, t.vertex1.z, 0};
        blob[L.offTriVert + 2 * pos + 1] = float4{t.vertex2.x, t.vertex2.y, t.vertex2.z, 0};
        reinterpret_cast<int*>(&blob[L.offTriPos])[i] = pos;
    }
    for (int i = 0; i < L.numMaterials; ++i) {
        const ptss_material& m = s.materials[i];
        float4* o = &blob[L.offMaterial + 5 * i];
        o[0] = float4{m.diffuseColor.x, m.diffuseColor.y, m.diffuseColor.z, m.diffAvg};
        o[1] = float4{m.specularColor.x, m.specularColor.y, m.specularColor.z, m.specAvg};
        o[2] = float4{m.absorption.x, m.absorption.y, m.absorption.z, m.refrAvg};
        o[3] = float4{m.emmitance.x, m.emmitance.y, m.emmitance.z, m.roughness};
        o[4] = float4{m.specularExponent, m.indexOfRefraction, u2f((uint32_t)(unsigned char)m.flags), 0};
    }
    for (int i = 0; i < L.numPointLights; ++i) {
        const ptss_point_light& p = s.pointLights[i];
        blob[L.offPointLight + 2 * i + 0] = float4{p.position.x, p.position.y, p.position.z, 0};
        blob[L.offPointLight + 2 * i + 1] = float4{p.power.x, p.power.y, p.power.z, 0};
    }
    for (int i = 0; i < L.numAreaLights; ++i) {
        const ptss_area_light& a = s.areaLights[i];
        // getAreaLightPoint picks triangle triangleIdx or triangleIdx + 1 (CudaTracer.cu:408): both as stored positions
        const int* triPos = reinterpret_cast<const int*>(&blob[L.offTriPos]);
        blob[L.offAreaLight + 2 * i] = float4{a.power.x, a.power.y, a.power.z, u2f((uint32_t)triPos[a.triangleIdx])};
        blob[L.offAreaLight + 2 * i + 1] = float4{u2f((uint32_t)triPos[a.triangleIdx + 1]), 0, 0, 0};
    }
}

int validateScene(const ptss_scene_desc& s) {
    if ((s.numSpheres && !s.spheres) || (s.numTriangles && !s.triangles) || (s.numMaterials && !s.materials) ||
        (s.numPointLights && !s.pointLights) || (s.numAreaLights && !s.areaLights))
        return fail(PTSS_EINVAL, "scene: null array with non-zero count");
    for (size_t i = 0; i < s.numSpheres; ++i)
        if (s.spheres[i].materialIdx < 0 || (size_t)s.spheres[i].materialIdx >= s.numMaterials)
            return fail(PTSS_EINVAL, "scene: sphere materialIdx out of range");
    for (size_t i = 0; i < s.numTriangles; ++i)
        if (s.triangles[i].materialIdx < 0 || (size_t)s.triangles[i].materialIdx >= s.numMaterials)
            return fail(PTSS_EINVAL, "scene: triangle materialIdx out of range");
    for (size_t i = 0; i < s.numAreaLights; ++i)
        if (s.areaLights[i].triangleIdx < 0 || (size_t)s.areaLights[i].triangleIdx + 1 >= s.numTriangles)
            return fail(PTSS_EINVAL, "scene: area light needs triangles [triangleIdx, triangleIdx+1]");
    return PTSS_OK;
}

ptss::FrameBuffers frameBuffers(const ptss_context* c, int laneIdx, ptss_uchar4* pixels, int sample) {
    const Lane& ln = c->lanes[(size_t)laneIdx];
    ptss::FrameBuffers fb{};
    fb.pool[0] = ln.dPool[0];
    fb.pool[1] = ln.dPool[1];
    fb.rngHome = c->dRngHome;
    fb.counts = ln.dCounts[c->countParity];
    fb.countsNext = ln.dCounts[1 - c->countParity];
    fb.shardCount0 = ln.dShardCount0;
    fb.lastCounts = ln.dLastCounts;
    fb.totalRayBounces = c->dTotal + laneIdx;
    fb.guardTimeouts = reinterpret_cast<uint32_t*>(c->dTotal + ptss::kMaxLanes + 8);
    fb.accum = c->dAccum;
    fb.fsum = c->dFsum;
    fb.staged = c->dStaged ? c->dStaged + (c->stagedTwice ? (size_t)(c->frameIndex & 1u) * c->capacity * c->samples : 0) : nullptr;
    fb.quantTable = reinterpret_cast<const float*>(c->dScene + c->layout.offQuant);
    fb.pixels = pixels;
    fb.regionCap = ln.regionCap;
    fb.numPixels = c->numPixels;
    fb.plane = c->capacity;
    fb.samples = c->samples;
    fb.firstTiles = c->samples * (c->capacity / ptss::kBlock);
    // The reference stops bouncing once <= 128 rays are live IN THE WHOLE FRAME (CudaTracer.cu:622). A
    // shard of a multi-GPU frame cannot know the frame-wide count without a collective per bounce, so a context with
    // tileWorld > 1 never stops early; the two agree whenever the frame-wide count stays above 128. (The lanes of ONE
    // context do know each other's counts: the guard is exact for any number of lanes.)
    fb.minLive = c->tile.world > 1 ? 0u : ptss::kMinLiveRays;
    fb.inverseTicks = 1.f / (float)((int)c->samples * (sample + 1));  // CudaTracer.cu:94 (S = 1: 1.f / (ticks + 1))
    fb.defaultColor[0] = c->defaultColor[0];
    fb.defaultColor[1] = c->defaultColor[1];
    fb.defaultColor[2] = c->defaultColor[2];
    fb.laneIndex = (uint32_t)laneIdx;
    fb.laneCount = (uint32_t)c->lanes.size();
    fb.frameRays = c->numPixels * c->samples;
    fb.numPeers = fb.laneCount - 1;
    fb.myDone = ln.dDone;
    fb.myFrameDone = ln.dDone + ptss::kCountWords;
    fb.frameSeq = c->frameIndex;
    fb.joinsFrame = (laneIdx + 1 == (int)c->lanes.size()) ? 1u : 0u;
    uint32_t p = 0;
    for (size_t k = 0; k < c->lanes.size(); ++k) {
        if ((int)k == laneIdx) continue;
        fb.peerCounts[p] = c->lanes[k].dCounts[c->countParity];
        fb.peerDone[p] = c->lanes[k].dDone;
        fb.peerFrameDone[p] = c->lanes[k].dDone + ptss::kCountWords;
        ++p;
    }
    return fb;
}

void drainKernelEvents(ptss_context* c, bool wait) {
    size_t k = 0;
    for (size_t i = 0; i < c->evBusy.size(); ++i) {
        EventPair p = c->evBusy[i];
        hipError_t q = wait ? hipEventSynchronize(p.b) : hipEventQuery(p.b);
        if (q == hipSuccess) {
            float start = 0, ms = 0;   // start since the epoch (coarse at large values), duration from the pair itself (exact)
            if (hipEventElapsedTime(&start, c->evEpoch, p.a) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
            {
                if (c->kernelSpans.size() >= (1u << 20)) c->kernelSpans.clear();   // nobody is asking (ptss_bounce_kernel_time empties it)
                c->kernelSpans.emplace_back((double)start, (double)start + (double)ms);
            }
            c->evFree.push_back(p);
        } else {
            c->evBusy[k++] = p;
        }
    }
    c->evBusy.resize(k);
    (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
}

// Frame lanes: did a lane give up waiting for a peer since the last check (FrameBuffers::guardTimeouts)? Called by the
// entry points that have just synchronised; one 4-byte read-back, and only in contexts with more than one lane.
int checkLaneTimeouts(ptss_context* c) {
    if (c->lanes.size() < 2 && !c->oneLaunch && !c->oneLaunchAlt) return PTSS_OK;   // only kernels that wait for others can time out
    uint32_t v = 0;
    HIP_TRY(hipMemcpy(&v, c->dTotal + ptss::kMaxLanes + 8, sizeof(v), hipMemcpyDeviceToHost));
    if (v != c->timeoutsSeen) {
        c->timeoutsSeen = v;
        return fail(PTSS_ETIMEOUT, "a bounded wait on the device expired (a frame lane for a peer lane, or a workgroup of the one-launch "
                                   "frame kernel for its shard): the frame was not traced as specified");
    }
    return PTSS_OK;
}

}  // namespace

extern "C" {

int ptss_version(void) { return PTSS_VERSION; }

const char* ptss_error_string(int code) {
    switch (code) {
        case PTSS_OK: return "ok";
        case PTSS_EINVAL: return "invalid argument";
        case PTSS_EHIP: return "HIP runtime error";
        case PTSS_ENODEVICE: return "no usable HIP device";
        case PTSS_ENOMEM: return "out of memory";
        case PTSS_ERANGE: return "buffer too small or index out of range";
        case PTSS_ETIMEOUT: return "a frame lane timed out waiting for a peer lane";
        default: return "unknown error";
    }
}

const char* ptss_last_error_detail(void) { return g_detail.c_str(); }

int ptss_default_config(ptss_render_config* cfg) {
    if (!cfg) return fail(PTSS_EINVAL, "cfg is null");
    memset(cfg, 0, sizeof(*cfg));
    cfg->structSize = (unsigned int)sizeof(*cfg);
    cfg->width = 512;  // DIM, CudaUtils.h:7
    cfg->height = 512;
    cfg->seed = 0x5EEDull;
    cfg->maxIterations = 15;  // CudaTracer.h:39
    cfg->device = 0;
    cfg->tileRank = 0;
    cfg->tileWorld = 1;
    cfg->bandRows = 8;
    cfg->syncEachFrame = 1;
    cfg->floatAccumulator = 0;
    cfg->timeKernels = 0;
    cfg->samplesPerPass = 1;
    cfg->everySphereLoop = 0;
    cfg->frameLanes = 0;
    cfg->lanesFreeRun = 0;
    cfg->oneLaunchFrames = 0;
    return PTSS_OK;
}

int ptss_create(const ptss_scene_desc* scene, const ptss_render_config* cfg, ptss_context** out) {
    if (!scene || !cfg || !out) return fail(PTSS_EINVAL, "null argument");
    if (cfg->structSize != (unsigned int)sizeof(ptss_render_config))
        return fail(PTSS_EINVAL, "cfg->structSize is not this library's sizeof(ptss_render_config): the caller was built against another "
                                 "ptss.h (compare ptss_version() with PTSS_VERSION) or did not start from ptss_default_config");
    if (cfg->width <= 0 || cfg->height <= 0 || (long long)cfg->width * cfg->height > (1ll << 31) - 256)
        return fail(PTSS_EINVAL, "bad frame size");
    if (cfg->maxIterations == 0 || cfg->maxIterations > (unsigned)ptss::kMaxBounces)
        return fail(PTSS_EINVAL, "maxIterations must be in [1, 64]");
    if (cfg->tileWorld <= 0 || cfg->tileRank < 0 || cfg->tileRank >= cfg->tileWorld || cfg->bandRows <= 0)
        return fail(PTSS_EINVAL, "bad tile spec");
    const int spp = cfg->samplesPerPass == 0 ? 1 : cfg->samplesPerPass;
    if (spp < 1 || spp > 64) return fail(PTSS_EINVAL, "samplesPerPass must be in [1, 64]");
    if ((long long)cfg->width * cfg->height >= (1ll << 26)) return fail(PTSS_EINVAL, "frame too large (>= 2^26 pixels)");
    {
        // A ray's word inside its shard's region is addressed in 32 bits (slotWord: 19 planes per block): the rays of one
        // pass — local pixels x sample lanes, rounded up to whole tiles per shard — must stay below 2^32 / 19 (~226
        // million; the whole pool may be larger than 4 GB, regions are based with 64-bit arithmetic).
        long long rows = 0;
        for (int y = 0; y < cfg->height; ++y)
            if ((y / cfg->bandRows) % cfg->tileWorld == cfg->tileRank) ++rows;
        const unsigned long long rays = ((unsigned long long)cfg->width * rows + 255ull) / 256ull * 256ull * (unsigned long long)spp;
        if ((rays + (unsigned long long)ptss::kShards * ptss::kBlock) * ptss::kRayPlanes >= (1ull << 32))
            return fail(PTSS_EINVAL, "too many rays per pass: width x local rows x samplesPerPass must stay below ~226 million");
    }
    int rc = validateScene(*scene);
    if (rc != PTSS_OK) return rc;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(PTSS_ENODEVICE, "hipGetDeviceCount found no device (libptss has no CPU path)", e);
    }
    if (cfg->device < 0 || cfg->device >= ndev) return fail(PTSS_ENODEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));

    ptss_context* c = new (std::nothrow) ptss_context();
    if (!c) return fail(PTSS_ENOMEM, "context");
    c->cfg = *cfg;
    c->maxIterations = cfg->maxIterations;
    c->samples = (uint32_t)spp;
    c->defaultColor[0] = scene->defaultColor.x;
    c->defaultColor[1] = scene->defaultColor.y;
    c->defaultColor[2] = scene->defaultColor.z;
    // Camera(), RenderStructs.h:51-52
    c->camera.rotation = q4(1, 0, 0, 0);
    c->camera.position = v3(0, 0, 0);
    c->camera.zNear = -0.1f;
    c->camera.zFar = -100.0f;
    c->camera.fieldOfView = ptm::kPi / 2.0f;

    int localRows = 0;
    for (int y = 0; y < cfg->height; ++y)
        if ((y / cfg->bandRows) % cfg->tileWorld == cfg->tileRank) ++localRows;
    c->tile = ptss::TileMap{cfg->width, cfg->height, localRows, cfg->tileRank, cfg->tileWorld, cfg->bandRows};
    c->numPixels = (uint32_t)cfg->width * (uint32_t)localRows;
    const uint32_t gran = ptss::kBlock > 256 ? (uint32_t)ptss::kBlock : 256u;  // pixel planes are whole tiles
    c->capacity = ((c->numPixels + gran - 1) / gran) * gran;
    if (c->capacity == 0) c->capacity = gran;
    // Frame lanes (ptss_device.h): cfg.frameLanes, or by the size of a pass when 0 — launch-shaped passes (a few
    // resident rounds per launch) gain a fifth from a second lane, wide ones nothing.
    int numLanes = cfg->frameLanes;
    if (numLanes < 0 || numLanes > ptss::kMaxLanes) return (delete c, fail(PTSS_EINVAL, "frameLanes must be in [0, 4]"));
    {
        const unsigned long long rays = (unsigned long long)c->numPixels * c->samples;
        // measured (tools/lanes_bench.py / tools/s1_modes.py, Mrays/s with 1 / 2 lanes, one sample per tick, with the lanes coupled on
        // the device — flushKernel's finished-frames counters): 1920x1080 (2.1 M rays per pass) 13,800 / 16,300; 3840x2160, 1,024 spheres
        // (8.3 M) 4,110 / 4,360 (three lanes 4,580, four 3,470); 1280x720 (0.92 M) 10,010 / 11,160; 1024x576 (0.59 M) 8,050 / 9,090;
        // 800x600 (0.48 M) 7,490 / 7,900; 640x480 (0.31 M) 5,480 / 5,380; 512x512 (0.26 M) 4,730 / 4,570 — below ~0.2 ms a pass is ten launch
        // latencies, nothing to overlap; 1920x1080 S = 40 (83 M) 17,760 / 17,750. Four lanes (five streams with the caller's) share
        // hardware queues and serialise. (With the lanes coupled through stream events the gain at 1280x720 was inside the noise.)
        // All of that is the FREE-RUNNING mode (cfg.lanesFreeRun); ordered strictly on the caller's stream — a fork and a join per
        // frame — two lanes ran at 13,180 against one lane's 13,800, so the library's own choice is then one lane.
        // Round 3, wide passes again (tools/lanes_large.py, interleaved, twice): 3840x2160 S = 4, 12 bounces, 1,024 spheres (33 M rays per
        // pass, the late bounces a hundredth of that) 5,958-5,970 / 6,077; 1920x1080 S = 40 (83 M) 18,273-18,294 / 18,420-18,454: a second
        // lane is worth +2.0 % / +0.8 % there. The rule stays at 2^24 all the same: a caller who wants it asks for frameLanes = 2; by
        // default a wide pass keeps one lane, one set of pools (the second doubles 5-16 GB) and launches that do not overlap, so that
        // a kernel's duration means the same in a profile and in ptss_bounce_kernel_time.
        if (numLanes == 0) numLanes = (cfg->lanesFreeRun && rays >= (3ull << 17) && rays <= (1ull << 24)) ? 2 : 1;
    }
    c->lanes.resize((size_t)numLanes);
    uint32_t shardCount0[ptss::kMaxLanes][ptss::kShards] = {{0}};
    {
        // bounce 0 walks S sample planes of `capacity` pixels (capacity = numPixels rounded up to a tile); tile t belongs
        // to shard t % kShards and to round t / kShards, round R to lane R % numLanes
        const uint32_t tilesPerPlane = c->capacity / ptss::kBlock;
        const uint32_t tiles = tilesPerPlane * c->samples;
        const uint32_t rounds = (tiles + ptss::kShards - 1) / ptss::kShards;
        for (int k = 0; k < numLanes; ++k) {
            const uint32_t laneRounds = (rounds + (uint32_t)numLanes - 1 - (uint32_t)k) / (uint32_t)numLanes;
            c->lanes[(size_t)k].regionCap = (laneRounds ? laneRounds : 1) * ptss::kBlock;
            c->lanes[(size_t)k].maxBlocks = (int)(c->lanes[(size_t)k].regionCap / ptss::kBlock) * ptss::kShards;
        }
        for (uint32_t t = 0; t < tiles; ++t) {
            const uint32_t first = (t % tilesPerPlane) * ptss::kBlock;  // first pixel of the tile inside its plane
            uint32_t cnt = 0;
            if (first < c->numPixels) cnt = c->numPixels - first < (uint32_t)ptss::kBlock ? c->numPixels - first : (uint32_t)ptss::kBlock;
            shardCount0[(t / ptss::kShards) % (uint32_t)numLanes][t % ptss::kShards] += cnt;
        }
    }

    // Scenes with many spheres get the chunked image (see accelEligible / packScene); cfg.everySphereLoop keeps the plain one
    const bool wantAccel = accelEligible(*scene) && !cfg->everySphereLoop;
    std::vector<float4> blob, blobAlt;
    packScene(*scene, c->layout, blob, wantAccel);
    if (wantAccel) packScene(*scene, c->layoutAlt, blobAlt, false);
    c->haveAccel = c->accelActive = wantAccel;
    // Scenes whose image fits the default 64 KiB dynamic-LDS window are staged in LDS; larger ones are read in place
    // (wave-uniform scalar loads + per-lane gathers from global memory) — same kernel, same results, no size limit.
    const bool sceneFitsLds = ptss::bounceLdsBytes(c->layout, true) <= 64 * 1024;
    const bool sceneFitsLdsAlt = wantAccel && ptss::bounceLdsBytes(c->layoutAlt, true) <= 64 * 1024;

#define CREATE_TRY(expr)                                  \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) {                           \
            int _rc = fail(_e == hipErrorOutOfMemory ? PTSS_ENOMEM : PTSS_EHIP, #expr, _e); \
            ptss_destroy(c);                              \
            return _rc;                                   \
        }                                                 \
    } while (0)

    CREATE_TRY(hipMalloc(&c->dScene, blob.size() * sizeof(float4)));
    CREATE_TRY(hipMemcpy(c->dScene, blob.data(), blob.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (wantAccel) {
        CREATE_TRY(hipMalloc(&c->dSceneAlt, blobAlt.size() * sizeof(float4)));
        CREATE_TRY(hipMemcpy(c->dSceneAlt, blobAlt.data(), blobAlt.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMalloc(&c->dRngHome, (size_t)ptss::kHomeWords * c->capacity * c->samples * sizeof(uint32_t)));
    for (int k = 0; k < numLanes; ++k) {
        Lane& ln = c->lanes[(size_t)k];
        const size_t poolBytes = (size_t)ptss::kRayPlanes * ln.regionCap * ptss::kShards * sizeof(float);
        CREATE_TRY(hipMalloc(&ln.dPool[0], poolBytes));
        CREATE_TRY(hipMalloc(&ln.dPool[1], poolBytes));
        for (int b = 0; b < 2; ++b) {
            CREATE_TRY(hipMalloc(&ln.dCounts[b], ptss::kCountWords * sizeof(uint32_t)));
            CREATE_TRY(hipMemset(ln.dCounts[b], 0, ptss::kCountWords * sizeof(uint32_t)));
        }
        CREATE_TRY(hipMalloc(&ln.dLastCounts, ptss::kCountWords * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(ln.dLastCounts, 0, ptss::kCountWords * sizeof(uint32_t)));
        for (int s = 0; s < ptss::kShards; ++s)  // arm bounce 0 of the first frame (flushKernel arms every later one)
            CREATE_TRY(hipMemcpy(ln.dCounts[0] + ptss::countIndex(0, s), &shardCount0[k][s], sizeof(uint32_t), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&ln.dShardCount0, sizeof(shardCount0[k])));
        CREATE_TRY(hipMemcpy(ln.dShardCount0, shardCount0[k], sizeof(shardCount0[k]), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&ln.dDone, (ptss::kCountWords + ptss::kCountStride) * sizeof(uint32_t)));  // + the finished-frames counter
        CREATE_TRY(hipMemset(ln.dDone, 0, (ptss::kCountWords + ptss::kCountStride) * sizeof(uint32_t)));
        if (numLanes > 1) {
            CREATE_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            CREATE_TRY(hipEventCreateWithFlags(&ln.evDone[0], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&ln.evDone[1], hipEventDisableTiming));
        }
        CREATE_TRY(hipHostMalloc(&ln.hCounts, 4 * ptss::kCountWords * sizeof(uint32_t), hipHostMallocDefault));
        for (int q = 0; q < 4; ++q) CREATE_TRY(hipEventCreateWithFlags(&ln.hintEvent[q], hipEventDisableTiming));
    }
    if (numLanes > 1) CREATE_TRY(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    CREATE_TRY(hipMalloc(&c->dTotal, kTotalWords * sizeof(unsigned long long)));
    CREATE_TRY(hipMemset(c->dTotal, 0, kTotalWords * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(&c->dAccumOwned, (size_t)3 * c->capacity * sizeof(uint32_t)));
    CREATE_TRY(hipMemset(c->dAccumOwned, 0, (size_t)3 * c->capacity * sizeof(uint32_t)));
    c->dAccum = c->dAccumOwned;
    if (cfg->floatAccumulator) {
        CREATE_TRY(hipMalloc(&c->dFsum, (size_t)3 * c->capacity * c->samples * sizeof(float)));
        CREATE_TRY(hipMemset(c->dFsum, 0, (size_t)3 * c->capacity * c->samples * sizeof(float)));
    }
    if (c->samples > 1) {
        c->stagedTwice = numLanes > 1 && cfg->lanesFreeRun != 0;
        const size_t words = (size_t)c->capacity * c->samples * (c->stagedTwice ? 2 : 1);
        CREATE_TRY(hipMalloc(&c->dStaged, words * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(c->dStaged, 0, words * sizeof(uint32_t)));
        if (c->stagedTwice)
            for (int q = 0; q < 2; ++q) CREATE_TRY(hipEventCreateWithFlags(&c->evDisplay[q], hipEventDisableTiming));
    }
    CREATE_TRY(hipEventCreate(&c->evStart));
    CREATE_TRY(hipEventCreate(&c->evStop));
    CREATE_TRY(hipEventCreate(&c->evEpoch));

    // curandSetupKernel (CudaTracer.cu:722-724): per-pixel subsequence via the 2^67 jump table
    {
        std::vector<uint32_t> table(ptrng::kJumpTableWords);
        ptrng::build_subsequence_table(table.data());
        uint32_t* dTable = nullptr;
        CREATE_TRY(hipMalloc(&dTable, table.size() * sizeof(uint32_t)));
        hipError_t e1 = hipMemcpy(dTable, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        hipError_t e2 = e1;
        if (e1 == hipSuccess && c->numPixels > 0)
            e2 = ptss::launchRngInit(nullptr, c->dRngHome, c->capacity, c->samples, c->tile, cfg->seed, dTable);
        hipError_t e3 = e2 == hipSuccess ? hipEventRecord(c->evEpoch, nullptr) : e2;
        if (e3 == hipSuccess) e3 = hipDeviceSynchronize();
        (void)hipFree(dTable);
        CREATE_TRY(e3);
    }

    // Scene access path: staged into LDS whenever the image fits (north_star); larger scenes are read in place. (On the
    // 38-primitive "mixed" scene reading in place — wave-uniform s_load through the scalar cache — measured 16 % slower,
    // profiles/README.md r01.)
    c->sceneInLds = sceneFitsLds;
    c->sceneInLdsAlt = sceneFitsLdsAlt;
    {
        // Launches wider than 16 resident rounds stop growing: a workgroup then walks several tiles and stages the scene
        // into LDS once for all of them. One round = CUs x workgroups per CU of THIS scene's bounce kernel (LDS image and
        // register budget decide: 7 for the 38-primitive scenes, 3-4 for the many-sphere image), so 16 rounds =
        // CUs x perCU workgroups per shard (kShards = 16 shards). Measured on the mixed scene (256 x 7 = 1,792 per shard):
        // 448: -4.6 %, 896: -1.3 %, 1,280-3,584: equal, uncapped: -2 % (profiles/README.md).
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, cfg->device));
        const int perCU = ptss::bounceOccupancyBlocksPerCU(c->layout, c->sceneInLds, wantAccel);
        c->gridCap = prop.multiProcessorCount * (perCU > 0 ? perCU : 4) * 16 / ptss::kShards;
        // One launch per frame (ptss_kernels.hip frameKernel): only when every workgroup of the frame's grid is resident at once —
        // its workgroups wait for each other — i.e. bounce-0 tiles <= CUs x resident workgroups per CU of THAT kernel with this
        // scene's LDS image. The occupancy API over-reports by one workgroup per CU for kernels of more than 96 SGPRs
        // (MI355X_MICROARCH.md, "Residency and cooperative launch"): one is kept in reserve. One lane, scene staged in LDS.
        auto qualifies = [&](const ptss::SceneLayout& lay, bool inLds) {
            if (cfg->oneLaunchFrames <= 0 || numLanes != 1 || !inLds) return false;   // opt-in (include/ptss.h)
            const int resident = ptss::frameOccupancyBlocksPerCU(lay, lay.sphereBounded != 0) - 1;
            return resident >= 1 && c->lanes[0].maxBlocks <= prop.multiProcessorCount * resident;
        };
        c->oneLaunch = qualifies(c->layout, c->sceneInLds);
        c->oneLaunchAlt = wantAccel && qualifies(c->layoutAlt, c->sceneInLdsAlt);
#ifdef PTSS_TUNING_KNOBS   // measurement builds only (tools/build_variants.py "knobs"); the shipped library reads no environment
        if (const char* e = getenv("PTSS_SCENE_PATH")) {
            if (!strcmp(e, "scalar")) c->sceneInLds = c->sceneInLdsAlt = false;
        }
        if (const char* e = getenv("PTSS_GRID_CAP")) c->gridCap = atoi(e);
#endif
    }
#undef CREATE_TRY

    *out = c;
    return PTSS_OK;
}

int ptss_destroy(ptss_context* c) {
    if (!c) return PTSS_OK;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    drainKernelEvents(c, true);
    for (EventPair& p : c->evFree) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (Lane& ln : c->lanes) {
        for (int q = 0; q < 4; ++q)
            if (ln.hintEvent[q]) (void)hipEventDestroy(ln.hintEvent[q]);
        if (ln.hCounts) (void)hipHostFree(ln.hCounts);
        (void)hipFree(ln.dDone);
        if (ln.evDone[0]) (void)hipEventDestroy(ln.evDone[0]);
        if (ln.evDone[1]) (void)hipEventDestroy(ln.evDone[1]);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
        (void)hipFree(ln.dPool[0]);
        (void)hipFree(ln.dPool[1]);
        (void)hipFree(ln.dCounts[0]);
        (void)hipFree(ln.dCounts[1]);
        (void)hipFree(ln.dShardCount0);
        (void)hipFree(ln.dLastCounts);
    }
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    for (int q = 0; q < 2; ++q)
        if (c->evDisplay[q]) (void)hipEventDestroy(c->evDisplay[q]);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->evEpoch) (void)hipEventDestroy(c->evEpoch);
    (void)hipFree(c->dScene);
    (void)hipFree(c->dSceneAlt);
    (void)hipFree(c->dRngHome);
    (void)hipFree(c->dTotal);
    (void)hipFree(c->dAccumOwned);
    (void)hipFree(c->dFsum);
    (void)hipFree(c->dStaged);
    delete c;
    return PTSS_OK;
}

int ptss_generate_frame(ptss_context* c, ptss_uchar4* pixels, int ticks) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    hipStream_t st = c->stream;
    HIP_TRY(hipSetDevice(c->cfg.device));  // contexts of several devices may live in one thread
    c->lastTicks = ticks;
    if (c->numPixels == 0) {  // a rank whose tile is empty (more ranks than row bands): nothing to render
        if (c->resetTicksThisFrame) c->lastResetTick = ticks;
        c->resetTicksThisFrame = false;
        c->lastMs = 0.0f;
        return PTSS_OK;
    }

    if (c->haveAccel) {  // the chunked image assumes a camera within the geometry's magnitude range (packScene)
        const bool want = cameraInRange(c->camera);
        if (want != c->accelActive) {
            std::swap(c->dScene, c->dSceneAlt);
            std::swap(c->layout, c->layoutAlt);
            std::swap(c->sceneInLds, c->sceneInLdsAlt);
            std::swap(c->oneLaunch, c->oneLaunchAlt);
            c->accelActive = want;
            c->cameraDirty = true;
        }
    }
    bool forkNeeded = false;
    if (c->resetTicksThisFrame) {  // CudaTracer.cu:602-608
        forkNeeded = true;
        c->lastResetTick = ticks;
        HIP_TRY(ptss::launchClear(st, frameBuffers(c, 0, pixels, 0)));
        c->resetTicksThisFrame = false;
    }
    const int sample = ticks - c->lastResetTick;
    const int K = (int)c->lanes.size();
    ptss::FrameBuffers fbs[ptss::kMaxLanes];
    for (int k = 0; k < K; ++k) fbs[k] = frameBuffers(c, k, pixels, sample);

    if (c->cfg.syncEachFrame) HIP_TRY(hipEventRecord(c->evStart, st));  // :611

    const int numIterations = c->usePathTracer ? (int)c->maxIterations : 1;  // :620

    ptss::EyeParams eye;
    eye.camera = c->camera;
    eye.s = -2 * ptm::tan(c->camera.fieldOfView * 0.5f);  // :334
    eye.aspect = (float)c->tile.height / (float)c->tile.width;
    eye.invW = 1.0f / c->tile.width;
    eye.invH = 1.0f / c->tile.height;
    if (c->cameraDirty) {  // origin-only parts of the primary-ray tests (computeEyeRaysKernel :614 itself is fused into bounce 0)
        forkNeeded = true;
        HIP_TRY(ptss::launchPrimaryPrep(st, c->dScene, c->layout, c->camera.position));
        c->cameraDirty = false;
    }

    // harvest the newest finished live-count readbacks (never blocks)
    for (Lane& ln : c->lanes)
        for (int q = 0; q < 4; ++q) {
            if (!ln.hintPending[q]) continue;
            if (hipEventQuery(ln.hintEvent[q]) == hipSuccess) {
                const uint32_t* src = ln.hCounts + (size_t)q * ptss::kCountWords;
                for (int b = 0; b <= ptss::kMaxBounces; ++b) {
                    uint32_t mx = 0;
                    for (int s = 0; s < ptss::kShards; ++s) {
                        const uint32_t v = src[ptss::countIndex(b, s)];
                        if (v > mx) mx = v;
                    }
                    ln.hint[b] = mx;
                }
                ln.haveHint = true;
                ln.hintPending[q] = false;
            }
        }
    (void)hipGetLastError();

    // Several lanes: each runs on its own stream, forked from the caller's here and joined into it below; their
    // launches are issued round-robin, bounce by bounce, so that the streams advance together.
    // STRICT ordering (the default): every frame forks, so the lanes start behind whatever the caller put on its stream
    // before this call — a reader of the previous frame's pixels or accumulator, a zero-fill, a newly bound buffer's writer.
    // FREE-RUNNING (cfg.lanesFreeRun, opt-in): the fork happens only when the library's own work on the caller's stream
    // requires it (the clear of a reset, the camera precomputes) — otherwise a lane's next frame depends on nothing but
    // its own previous one, and the lanes run on, frame after frame, while the caller's stream merely waits for each
    // frame's end (the join below); the caller has promised not to touch the buffers in between (include/ptss.h).
    if (K > 1 && (forkNeeded || c->frameIndex == 0 || !c->cfg.lanesFreeRun)) {
        HIP_TRY(hipEventRecord(c->evFork, st));
        for (Lane& ln : c->lanes) HIP_TRY(hipStreamWaitEvent(ln.stream, c->evFork, 0));
    }
    // Free-running lanes with S > 1: this frame parks its samples in the buffer of its parity, which displayKernel of the frame
    // two back (same parity, on the caller's stream) must have emptied — the only thing a free-running lane ever waits for
    // on the caller's side, and an event that has nearly always fired by now.
    if (c->stagedTwice && c->frameIndex >= 2)
        for (Lane& ln : c->lanes) HIP_TRY(hipStreamWaitEvent(ln.stream, c->evDisplay[c->frameIndex & 1u], 0));
    if (c->cfg.timeKernels) drainKernelEvents(c, false);
    // the shorter sphere candidate test: bounded geometry AND a camera within the same range (ray origins are the camera or points on primitives)
    const bool bounded = c->layout.sphereBounded != 0 && cameraInRange(c->camera);
    // The frame kernel's bounded form of the sphere test needs what the bounce kernels' needs (above); its grid is the frame's tiles.
    if (c->oneLaunch && K == 1) {   // every bounce in ONE launch (frameKernel): :622-633 without leaving the device
        Lane& ln = c->lanes[0];
        EventPair ev{nullptr, nullptr};
        if (c->cfg.timeKernels) {
            if (c->evFree.empty()) {
                if (c->evBusy.size() >= 4096) drainKernelEvents(c, true);
                if (c->evFree.empty()) {
                    HIP_TRY(hipEventCreate(&ev.a));
                    HIP_TRY(hipEventCreate(&ev.b));
                }
            }
            if (!ev.a) {
                ev = c->evFree.back();
                c->evFree.pop_back();
            }
            HIP_TRY(hipEventRecord(ev.a, st));
        }
        HIP_TRY(ptss::launchFrame(st, fbs[0], c->dScene, c->layout, numIterations, bounded, ln.maxBlocks, c->tile, eye));
        if (c->cfg.timeKernels) {
            HIP_TRY(hipEventRecord(ev.b, st));
            c->evBusy.push_back(ev);
        }
    } else
    for (int i = 0; i < numIterations; ++i) {  // :622-633, guard evaluated on the device
        for (int k = 0; k < K; ++k) {
            Lane& ln = c->lanes[(size_t)k];
            hipStream_t ls = K > 1 ? ln.stream : st;
            // grid: one tile per workgroup for the expected live count (+1.5 %), never more than the lane's share of the
            // frame; the kernel grid-strides, so a low hint costs time, not correctness
            int blocks = ln.maxBlocks;
            if (i > 0 && ln.haveHint) {
                // tiles for the fullest shard (+1.5 %), times kShards (workgroup b serves shard b % kShards)
                const unsigned long long tilesPerShard = ((unsigned long long)ln.hint[i] * 65 / 64 + ptss::kBlock) / ptss::kBlock + 1;
                const unsigned long long want = tilesPerShard * ptss::kShards;
                if (want < (unsigned long long)blocks) blocks = (int)want;
            }
            // launches wider than 16 resident rounds stop growing (gridCap, ptss_create)
            if (c->gridCap > 0 && c->gridCap * ptss::kShards < blocks) blocks = c->gridCap * ptss::kShards;
            EventPair ev{nullptr, nullptr};
            if (c->cfg.timeKernels) {
                if (c->evFree.empty()) {
                    if (c->evBusy.size() >= 4096) drainKernelEvents(c, true);
                    if (c->evFree.empty()) {
                        HIP_TRY(hipEventCreate(&ev.a));
                        HIP_TRY(hipEventCreate(&ev.b));
                    }
                }
                if (!ev.a) {
                    ev = c->evFree.back();
                    c->evFree.pop_back();
                }
                HIP_TRY(hipEventRecord(ev.a, ls));
            }
            // the peers' done totals once their bounce i - 1 of this frame has ended (all of those launches precede this
            // one in host order, so a kernel that waits for them never waits for something behind it in a shared queue)
            if (i > 0)
                for (int j = 0, p = 0; j < K; ++j)
                    if (j != k) fbs[k].peerTarget[p++] = c->lanes[(size_t)j].doneTarget[i - 1];
            HIP_TRY(ptss::launchBounce(ls, fbs[k], c->dScene, c->layout, i, i == numIterations - 1, c->sceneInLds, bounded, blocks, c->tile, eye));
            ln.doneTarget[i] += (uint32_t)blocks;  // every workgroup of the launch adds 1 to done[i][its shard] as it ends
            if (c->cfg.timeKernels) {
                HIP_TRY(hipEventRecord(ev.b, ls));
                c->evBusy.push_back(ev);
            }
        }
    }
    for (int k = 0; k < K; ++k) {
        Lane& ln = c->lanes[(size_t)k];
        hipStream_t ls = K > 1 ? ln.stream : st;
        // (flushKernel itself waits, on the device, until every peer lane has finished the previous frame: FrameBuffers::myFrameDone)
        ptss::FlushTargets targets{};
        for (int j = 0, p = 0; j < K; ++j)
            if (j != k) {
                for (int b = 0; b <= ptss::kMaxBounces; ++b) targets.target[p][b] = c->lanes[(size_t)j].doneTarget[b];
                ++p;
            }
        HIP_TRY(ptss::launchFlush(ls, fbs[k], numIterations, targets));  // :637
        // every 8th frame (and until a hint exists) copy counts[] to pinned memory for later grid sizing
        if (!ln.haveHint || (c->frameIndex & 7u) == 0) {
            const int q = (int)((c->frameIndex >> 3) & 3u);
            if (!ln.hintPending[q]) {
                HIP_TRY(hipMemcpyAsync(ln.hCounts + (size_t)q * ptss::kCountWords, ln.dLastCounts, ptss::kCountWords * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ls));
                HIP_TRY(hipEventRecord(ln.hintEvent[q], ls));
                ln.hintPending[q] = true;
            }
        }
        // the join: the caller's stream is ordered behind every lane's frame (the lanes themselves run on) — ONE event, behind
        // the last lane's flushKernel, which ends only when every other lane's has (FrameBuffers::joinsFrame); an event per lane cost 1-2 %
        if (K > 1 && k == K - 1) {
            HIP_TRY(hipEventRecord(ln.evDone[c->frameIndex & 1u], ls));
            HIP_TRY(hipStreamWaitEvent(st, ln.evDone[c->frameIndex & 1u], 0));
        }
    }
    if (c->samples > 1) {
        HIP_TRY(ptss::launchDisplay(st, fbs[0]));  // S > 1: add the pass's staged samples, then the display value
        if (c->stagedTwice) HIP_TRY(hipEventRecord(c->evDisplay[c->frameIndex & 1u], st));
    }
    c->countParity ^= 1;
    c->frameIndex++;

    if (c->cfg.syncEachFrame) {  // :639-642
        HIP_TRY(hipEventRecord(c->evStop, st));
        HIP_TRY(hipEventSynchronize(c->evStop));
        HIP_TRY(hipEventElapsedTime(&c->lastMs, c->evStart, c->evStop));
        // (a 4-byte read-back per frame: only where several lanes wait for each other; a one-launch context is checked by
        // ptss_synchronize and the ptss_read_* calls — its frames are a third of a millisecond)
        return c->lanes.size() > 1 ? checkLaneTimeouts(c) : PTSS_OK;
    }
    return PTSS_OK;
}

int ptss_set_camera(ptss_context* c, const ptss_camera* camera) {
    if (!c || !camera) return fail(PTSS_EINVAL, "null argument");
    c->camera = *camera;
    c->cameraDirty = true;
    c->resetTicksThisFrame = true;
    return PTSS_OK;
}

int ptss_get_camera(const ptss_context* c, ptss_camera* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->camera;
    return PTSS_OK;
}

int ptss_request_reset(ptss_context* c) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    c->resetTicksThisFrame = true;
    return PTSS_OK;
}

int ptss_set_mode(ptss_context* c, int usePathTracer) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    c->usePathTracer = usePathTracer != 0;
    c->resetTicksThisFrame = true;
    return PTSS_OK;
}

int ptss_set_max_iterations(ptss_context* c, unsigned int maxIterations) {
    if (!c || maxIterations == 0 || maxIterations > (unsigned)ptss::kMaxBounces)
        return fail(PTSS_EINVAL, "maxIterations must be in [1, 64]");
    c->maxIterations = maxIterations;
    return PTSS_OK;
}

int ptss_set_stream(ptss_context* c, void* hipStream) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    c->stream = (hipStream_t)hipStream;
    return PTSS_OK;
}

int ptss_bind_accumulator(ptss_context* c, uint32_t* dev) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    c->dAccum = dev ? dev : c->dAccumOwned;
    return PTSS_OK;
}

int ptss_accumulator_devptr(ptss_context* c, uint32_t** out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->dAccum;
    return PTSS_OK;
}

int ptss_float_accumulator_devptr(ptss_context* c, float** out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->dFsum;
    return PTSS_OK;
}

int ptss_alloc_pixels(ptss_context* c, ptss_uchar4** out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMalloc(out, (size_t)c->capacity * sizeof(ptss_uchar4)));
    HIP_TRY(hipMemset(*out, 0, (size_t)c->capacity * sizeof(ptss_uchar4)));
    return PTSS_OK;
}

int ptss_free_pixels(ptss_context* c, ptss_uchar4* dev) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    HIP_TRY(hipFree(dev));
    return PTSS_OK;
}

int ptss_local_pixels(const ptss_context* c, size_t* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->numPixels;
    return PTSS_OK;
}

int ptss_local_rows(const ptss_context* c, int* rows, int cap, int* count) {
    if (!c || !count) return fail(PTSS_EINVAL, "null argument");
    int n = 0;
    for (int y = 0; y < c->tile.height; ++y) {
        if ((y / c->tile.bandRows) % c->tile.world != c->tile.rank) continue;
        if (rows) {
            if (n >= cap) return fail(PTSS_ERANGE, "rows[] too small");
            rows[n] = y;
        }
        ++n;
    }
    *count = n;
    return PTSS_OK;
}

int ptss_synchronize(ptss_context* c) {
    if (!c) return fail(PTSS_EINVAL, "ctx is null");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return checkLaneTimeouts(c);
}

int ptss_read_accumulator(ptss_context* c, uint32_t* host, size_t count) {
    if (!c || !host) return fail(PTSS_EINVAL, "null argument");
    if (count != (size_t)3 * c->numPixels) return fail(PTSS_ERANGE, "count must be 3 * local pixels");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host, c->dAccum, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return checkLaneTimeouts(c);
}

int ptss_read_float_accumulator(ptss_context* c, float* host, size_t count) {
    if (!c || !host) return fail(PTSS_EINVAL, "null argument");
    if (!c->dFsum) return fail(PTSS_EINVAL, "context was created without floatAccumulator");
    if (count != (size_t)3 * c->numPixels) return fail(PTSS_ERANGE, "count must be 3 * local pixels");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->samples == 1) {
        HIP_TRY(hipMemcpy(host, c->dFsum, count * sizeof(float), hipMemcpyDeviceToHost));
    } else {  // per-stream sums, added in lane order 0..S-1 (a fixed order: reproducible, and what the oracle does)
        std::vector<float> lane(count);
        for (size_t k = 0; k < count; ++k) host[k] = 0.0f;
        for (uint32_t l = 0; l < c->samples; ++l) {
            HIP_TRY(hipMemcpy(lane.data(), c->dFsum + (size_t)3 * l * c->capacity, count * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < count; ++k) host[k] = host[k] + lane[k];
        }
    }
    return checkLaneTimeouts(c);
}

int ptss_read_pixels(ptss_context* c, const ptss_uchar4* dev, ptss_uchar4* host, size_t count) {
    if (!c || !dev || !host) return fail(PTSS_EINVAL, "null argument");
    if (count > c->numPixels) return fail(PTSS_ERANGE, "count exceeds local pixels");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host, dev, count * sizeof(ptss_uchar4), hipMemcpyDeviceToHost));
    return checkLaneTimeouts(c);
}

int ptss_read_rng_state(ptss_context* c, size_t local_pixel, uint32_t* out6) { return ptss_read_rng_state_lane(c, local_pixel, 0, out6); }

int ptss_read_rng_state_lane(ptss_context* c, size_t local_pixel, unsigned int lane, uint32_t* out6) {
    if (!c || !out6) return fail(PTSS_EINVAL, "null argument");
    if (local_pixel >= c->numPixels || lane >= c->samples) return fail(PTSS_ERANGE, "pixel or lane out of range");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out6, c->dRngHome + (size_t)ptss::kHomeWords * ((size_t)lane * c->capacity + local_pixel), 6 * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    return PTSS_OK;
}

int ptss_last_pass_ms(ptss_context* c, float* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->lastMs;
    return PTSS_OK;
}

int ptss_samples_since_reset(const ptss_context* c, int* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = c->resetTicksThisFrame ? 0 : (c->lastTicks - c->lastResetTick + 1) * (int)c->samples;
    return PTSS_OK;
}

int ptss_live_counts(ptss_context* c, uint32_t* out, int cap, int* n) {
    if (!c || !out || !n) return fail(PTSS_EINVAL, "null argument");
    const int numIterations = c->usePathTracer ? (int)c->maxIterations : 1;
    if (cap < numIterations) return fail(PTSS_ERANGE, "out[] too small");
    std::vector<uint32_t> raw(ptss::kCountWords), sum(ptss::kCountWords, 0u);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (const Lane& ln : c->lanes) {  // the frame's count = its lanes' counts added up
        HIP_TRY(hipMemcpy(raw.data(), ln.dLastCounts, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < raw.size(); ++k) sum[k] += raw[k];
    }
    // a bounce whose input is <= 128 rays did not run (CudaTracer.cu:622): report 0 from there on
    bool stopped = false;
    for (int i = 0; i < numIterations; ++i) {
        uint32_t total = 0;
        for (int s = 0; s < ptss::kShards; ++s) total += sum[ptss::countIndex(i, s)];
        if (total <= (c->tile.world > 1 ? 0u : ptss::kMinLiveRays)) stopped = true;
        out[i] = stopped ? 0u : total;
    }
    *n = numIterations;
    return checkLaneTimeouts(c);
}

int ptss_total_ray_bounces(ptss_context* c, unsigned long long* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long perLane[ptss::kMaxLanes];
    HIP_TRY(hipMemcpy(perLane, c->dTotal, sizeof(perLane), hipMemcpyDeviceToHost));
    *out = 0;
    for (size_t k = 0; k < c->lanes.size(); ++k) *out += perLane[k];
    return checkLaneTimeouts(c);
}

int ptss_guard_timeouts(ptss_context* c, unsigned int* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t v = 0;
    HIP_TRY(hipMemcpy(&v, c->dTotal + ptss::kMaxLanes + 8, sizeof(v), hipMemcpyDeviceToHost));
    *out = v;
    return PTSS_OK;
}

int ptss_one_launch_frames(const ptss_context* c, int* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = (c->oneLaunch && c->lanes.size() == 1) ? 1 : 0;
    return PTSS_OK;
}

int ptss_frame_lanes(const ptss_context* c, int* out) {
    if (!c || !out) return fail(PTSS_EINVAL, "null argument");
    *out = (int)c->lanes.size();
    return PTSS_OK;
}

int ptss_debug_counters(ptss_context* c, unsigned long long* out8) {
    if (!c || !out8) return fail(PTSS_EINVAL, "null argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
#if PTSS_DIAG
    HIP_TRY(ptss::readDiagCounters(out8));
#else
    for (int k = 0; k < 8; ++k) out8[k] = 0ull;   // the shipped library carries no counter (ptss_diag.h)
#endif
    return PTSS_OK;
}

int ptss_bounce_kernel_time(ptss_context* c, double* total_ms, unsigned long long* launches) {
    if (!c || !total_ms || !launches) return fail(PTSS_EINVAL, "null argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (const Lane& ln : c->lanes)
        if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
    drainKernelEvents(c, true);
    // union of the launch intervals (one lane: they do not overlap, and this is the sum of the durations)
    std::sort(c->kernelSpans.begin(), c->kernelSpans.end());
    double busy = 0.0, curA = 0.0, curB = -1.0;
    for (const auto& sp : c->kernelSpans) {
        if (curB < curA || sp.first > curB) {
            if (curB >= curA) busy += curB - curA;
            curA = sp.first;
            curB = sp.second;
        } else if (sp.second > curB) {
            curB = sp.second;
        }
    }
    if (curB >= curA) busy += curB - curA;
    *total_ms = busy;
    *launches = c->kernelSpans.size();
    c->kernelSpans.clear();
    HIP_TRY(hipEventRecord(c->evEpoch, c->stream));   // a fresh epoch for the next window: float32 ms stay fine-grained
    HIP_TRY(hipEventSynchronize(c->evEpoch));
    return checkLaneTimeouts(c);
}

}  // extern "C"
