// ptss_device.h — device-side data layout shared by the kernels (ptss_kernels.hip) and the
// context code (ptss_api.hip). See DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptmath.h"
#include "ptss_types.h"
#include "xorwow.h"

namespace ptss {

// ---- ray pools: struct-of-arrays in tile blocks ------------------------------------------------------------
// A pool is cut into kShards equal regions of regionCap slots; the workgroups with blockIdx % kShards == s read and write
// region s only and own the s-th live-ray counter, so the compaction atomics of one launch are spread over kShards
// addresses (one address sustains only ~88 returning atomics/us on MI355X — measured: a single counter was a 45 ps/ray
// serial floor, profiles/README.md). A region's survivors can never outnumber its input, so regionCap =
// ceil(tiles / kShards) * kBlock always suffices. Inside a region the rays of one tile (kBlock consecutive slots) form a
// BLOCK of kRayPlanes planes of kBlock words: word (slot, plane p) = ((slot / kBlock) * kRayPlanes + p) * kBlock +
// slot % kBlock. Lane i of a wave touches word i of a plane — 256-B contiguous wave accesses — and a plane's offset
// inside the block is a compile-time constant (addressing: ptss_kernels.hip, tileBlock / slotWord).
enum RayPlane : int {
    kOx = 0, kOy, kOz,        // origin
    kDx, kDy, kDz,            // direction
    kL0x, kL0y, kL0z,         // radiance0 (accumulated)
    kTx, kTy, kTz,            // radiance1 (throughput)
    kPix,                     // local pixel index (Ray::pixelOffset)
    kR0, kR1, kR2, kR3, kR4,  // XORWOW v[0..4]
    kRd,                      // XORWOW d
    kRayPlanes                // = 19 planes = 76 B per ray
};
constexpr int kHomeWords = 8;     // per-pixel RNG home record: v0..v4, d, 2 pad words (32 B)
constexpr int kMaxBounces = 64;  // counts has (kMaxBounces + 1) x kShards entries
// Build-time switches (eight; tools/build_variants.py builds A/B variants — the shipped library uses the defaults). What was
// tried and rejected with numbers lives in profiles/README.md and in the history, not behind switches here.
#ifndef PTSS_BLOCK
#define PTSS_BLOCK 256    // rays per tile = threads per workgroup (128-ray tiles: -20 % at one sample per tick, profiles/README.md)
#endif
#ifndef PTSS_SHARDS
#define PTSS_SHARDS 16    // pool regions / live-ray counters per bounce
#endif
#ifndef PTSS_CHUNK
#define PTSS_CHUNK 16     // spheres per chunk of the many-sphere traversal; with the kd-split order (ptss_api.hip spatialOrder),
                          // configs[4]'s scene at S = 4, same box: 4: 2,224, 8: 3,762, 16: 4,158-4,167, 32: 3,476 Mrays/s — every
                          // lane tests every chunk bound, so halving their number is worth more than the tighter fit of smaller chunks
#endif
#ifndef PTSS_MINWAVES
#define PTSS_MINWAVES 7   // __launch_bounds__ waves/SIMD of the unbounded-geometry instantiations: 72 VGPRs. Measured, same box:
                          // 5 waves (96 VGPRs) 13.3, 6 (80) 14.2, 7 (72) 14.5, 8 (64, spills) 11.4 Grays/s
#endif
#ifndef PTSS_MINWAVES_BOUNDED
#define PTSS_MINWAVES_BOUNDED 6   // the instantiations for bounded scenes (SceneLayout::sphereBounded: the shorter sphere test) want 80
                                  // registers: at 7 waves the shorter test costs 3 % (32 instead of 16 B of scratch), at 6 it gains — same-box
                                  // A/B against 7 waves + the long test: c3 +0.5 %, c2 +2.3 %, one sample per tick at 1080p +2.5-3 %
#endif
#ifndef PTSS_MINWAVES_FIRST
#define PTSS_MINWAVES_FIRST 6   // bounce 0's instantiation (eye rays fused in, camera-origin tests): at 7 waves it spills 32 B, at 6
                                // (80 VGPRs) none — same-box A/B: that kernel 3,035 -> 2,826 us per launch, the pass +0.9 %
#endif
#ifndef PTSS_ABLATE
#define PTSS_ABLATE 0   // measurement only (WRONG images): bit 0 no NEE, 1 no closest-hit loops, 2 no scatter, 3 no finishPath
#endif
#ifndef PTSS_DIAG
#define PTSS_DIAG 0     // diagnostic counters (ptss_diag.h; tools/*_hist.py, *_stat.py): bit 0 sphere candidates per lane, 1 scatter
                        // blocks, 2 chunk culling, 3 shadow-segment pairs, 4 shadow-queue lengths. 0: no counter exists in the code
#endif
constexpr int kBlock = PTSS_BLOCK;          // rays per tile = threads per workgroup
constexpr int kChunkSpheres = PTSS_CHUNK;
static_assert((kChunkSpheres & (kChunkSpheres - 1)) == 0, "chunk size must be a power of two");   // spheres per chunk of the many-sphere traversal
constexpr int kShards = PTSS_SHARDS;        // pool regions / live-ray counters per bounce
constexpr int kCountStride = 32;            // one counter per 128-B line
constexpr int kCountWords = (kMaxBounces + 1) * kShards * kCountStride;
__host__ __device__ inline int countIndex(int bounce, int shard) { return (bounce * kShards + shard) * kCountStride; }
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kMinLiveRays = 128;  // loop guard `numRays > 128`, CudaTracer.cu:622

// ---- scene blob: one contiguous array of float4 staged into LDS by every workgroup --------------
// (scene records are read by all lanes at the same index -> LDS broadcast reads)
struct SceneLayout {
    int numSpheres, numTriangles, numMaterials, numPointLights, numAreaLights;
    // offsets in float4 units
    int offSphere;      // S x {cx, cy, cz, radius^2}; with accelSpheres: the spheres in spatially sorted order, padded to whole
                        // chunks of kChunkSpheres with copies of the last one
    int offSphereMat;   // ceil(S/4) x 4 ints
    int offTri;         // T x 3: {v0.xyz, bits(materialIdx)}, {e1.xyz, bits(0xFFFFFFFE - original index)}, {e2.xyz, 0} — in storage order (triClassed)
    int offTriNormal;   // T x 3: {n0,0},{n1,0},{n2,0}
    int offTriVert;     // T x 2: {v1,0},{v2,0}   (area-light sampling)
    int offMaterial;    // M x 5: {diffuseColor, diffAvg},{specularColor, specAvg},{absorption, refrAvg},
                        //        {emmitance, roughness},{specularExponent, indexOfRefraction, bits(flags), 0}
    int offPointLight;  // P x 2: {position,0},{power,0}
    int offAreaLight;   // A x 2: {power, bits(position of triangle triangleIdx)}, {bits(position of triangleIdx + 1), 0, 0, 0}
    // Sphere acceleration (scenes with many spheres; packScene decides): chunks of kChunkSpheres consecutive sorted spheres
    // with a conservative bounding sphere each; a lane visits only the chunks its ray can touch (ptss_kernels.hip).
    int accelSpheres;   // 0: every sphere is tested by every ray (the reference's loop); 1: chunked
    int numChunks;
    int offChunk;       // numChunks x {Cx, Cy, Cz, inflated R^2}
    int offSphereOrig;  // ints: original (caller's) index of each sorted sphere — decides ties the way the reference's order does
    int offSpherePos;   // ints: the inverse, sorted position of each original index (regrouped traversal: winner by original index)
    int offQuant;       // 65 rows: the 8-bit tone-map thresholds T[0..256] (ptquant.h), read by finishPath
    int offPrimSphere;  // S x {o - centre, dot(v,v) - r^2}        written on the device per camera (primaryPrepKernel)
    int offPrimTri;     // T x 2: {o - v0, dot(e2, r)}, {r = cross(s, e1), 0}
    int offPrimChunk;   // many-sphere image: numChunks x {o - C, dot(v,v) - bound, rounded down}, per camera as well (bounce 0's chunk test)
    int totalVec4;
    int ldsVec4;        // rows [0, ldsVec4) are staged into LDS; the rest (the many-sphere integer tables: material, original
                        // index, position — read only when a hit is accepted) stay in global memory
    int neeSkipSafe;    // 1: light powers and diffuse colours are finite, so zero Lambert terms are exactly +-0
    int sphereBounded;  // 1: every |coordinate| <= 1e15 and every sphere radius in [1e-12, 1e15]: the sphere candidate tests may take
                        //    the two-instructions-shorter discriminant form (ptss_kernels.hip shiftInSphere<true>) while the camera is in range
    int neePairs;       // 1: at least two lights and at least four of five primitives reflect diffusely (diffAvg > 0): a lit point then
                        //    nearly always needs both of its shadow segments, and the kernels that test the pair together (shared origin
                        //    terms, pairAnyHit) pay: +4.8 % on configs[1]'s scene; with specular-only materials about the scene many entries
                        //    hold one segment and they do not: -1.4 % on configs[2]'s
    int triDetBounded;  // 1: every triangle has |e1| |e2| <= 2^100 (finite), so |det| = |e1 . (d x e2)| < 2^126 whenever
                        //    |d|^2 < 2^30 — the closest-hit triangle loop may then use the reciprocal's fast path unguarded
    int triClassed;     // 1: every vertex is finite (bounded geometry) and the triangles are STORED GROUPED BY EDGE CLASS (pttri.h; the
                        //    caller's order inside a group; T <= 255): the uniform triangle loops run one loop per class, each with the body that
                        //    leaves out the products with that class's exact-zero edge components; the closest hit decides by the key
                        //    (distance, ~original index), which is what the reference's sequential `dist <= distance` rule ends on.
                        //    0: the caller's order, the general body, the sequential rule
    uint32_t triClassPack[5];  // positions [begin(c), begin(c + 1)) hold the triangles of class code c = class(e1) * 4 + class(e2): the 17 begins
                               // (begin(16) = T) as BYTES, four per word — five scalar registers instead of seventeen (classed scenes have
                               // T <= 255); a loop header extracts its two bounds with two s_bfe (ptss_kernels.hip classBegin)
    int offTriPos;      // ints: stored position of each original triangle index
};

struct TileMap {
    int width, height;      // full frame
    int localRows;          // rows owned by this context
    int rank, world, bandRows;
};

struct EyeParams {  // computeEyeRay constants evaluated once on the host with ptm::tan
    ptss_camera camera;
    float s;        // -2 * tan(fov/2)
    float aspect;   // H / W
    float invW, invH;
};

// ---- frame lanes: one frame traced as K independent ray populations on K streams of one device -----------------
// A launch-shaped pass (one sample per pixel: ten kernels of 30-90 us) loses a fifth of its time to the ramp and tail of
// launches only a few resident rounds wide. Lane k owns the bounce-0 tiles of rounds R with R % K == k (round R = tiles
// 16 R .. 16 R + 15, one per shard), its own pools, counters and stream; all per-pixel state (random streams,
// accumulator, display) is shared — lanes touch disjoint pixels. The tail of one lane's launch overlaps the other lanes'
// kernels. The loop guard `numRays > 128` (CudaTracer.cu:622) stays a WHOLE-FRAME quantity: a lane whose own count is
// above 128 knows the frame's is; only a lane holding <= 128 rays (then at most one workgroup per shard has work) waits
// for its peers' counts of that bounce and adds them up. "Peer p's counts of bounce b are final" means: every workgroup of
// p's bounce b - 1 launch has ended. Each workgroup of a bounce kernel therefore adds 1 to its lane's done[b - 1][shard]
// as its last act (sixteen counters per bounce, one per 128-B line, never reset: they grow by the grid size of every
// frame's launch), and the host hands each launch the total the peers' counters reach once their bounce b - 1 of THIS
// frame is through (it knows every grid it launched). The waiter only ever depends on kernels that were enqueued before
// it — all lanes' bounce b - 1 launches precede any lane's bounce b in host order — so lanes that share a hardware queue
// cannot deadlock. (Earlier designs, measured: a stream-ordered hipStreamWriteValue32 after every kernel cost 9 %; a done
// word stored at the START of bounce b plus a one-thread signal kernel behind bounce b - 1 of lanes 1.. cost 4 %.)
// So the image is the one-lane image, exactly, for every K, whatever the streams' queue mapping.
constexpr int kMaxLanes = 4;

struct FrameBuffers {
    float* pool[2];          // ray pools (ping-pong), kRayPlanes planes each
    uint32_t* rngHome;       // kHomeWords words per local pixel: where a pixel's stream rests between paths
    uint32_t* counts;        // counts[countIndex(b, s)]: rays of shard s entering bounce b of the current frame (two buffers
                             // alternate per frame, so that a peer lane can still read this frame's counts after flushKernel)
    uint32_t* countsNext;    // the other buffer: flushKernel arms it for the next frame
    const uint32_t* shardCount0;  // [kShards] pixels per shard (constant per context): counts of bounce 0
    uint32_t* lastCounts;    // the previous frame's counts (copied by flushKernel before it re-arms `counts`)
    unsigned long long* totalRayBounces;
    uint32_t* accum;         // uint3 per local pixel (totalPixelColors)
    float* fsum;             // float3 per local pixel or nullptr
    const float* quantTable; // the same thresholds in global memory (flushKernel has no staged scene)
    uint32_t* staged;        // S > 1 only: this pass's sample of every stream, x | y << 8 | z << 16 (one plane per sample lane)
    ptss_uchar4* pixels;     // display buffer or nullptr
    uint32_t regionCap;      // slots per shard region (a multiple of kBlock); a region is regionCap * kRayPlanes words
    uint32_t numPixels;      // local pixels
    uint32_t plane;          // numPixels rounded up to kBlock: stride of the per-pixel planes (one plane per sample lane)
    uint32_t samples;        // S = cfg.samplesPerPass: independent random streams per pixel traced per pass
    uint32_t firstTiles;     // tiles of bounce 0 = S * plane / kBlock
    uint32_t minLive;        // a bounce runs while more than this many rays are live: 128 (CudaTracer.cu:622),
                             // 0 in a sharded context (the guard is a whole-frame quantity; DESIGN.md "Sharding")
    float inverseTicks;      // 1.f / (ticks - lastResetTick + 1)
    float defaultColor[3];
    // frame lanes (laneCount = 1: everything below is inert)
    uint32_t laneIndex, laneCount;
    uint32_t frameRays;              // rays all lanes together start a pass with (numPixels x samples): bounce 0's guard
    uint32_t numPeers;               // laneCount - 1
    const uint32_t* peerCounts[kMaxLanes - 1];  // the peers' counts[] of the current frame
    const uint32_t* peerDone[kMaxLanes - 1];    // the peers' done[countIndex(b, s)]: workgroups of shard s that ended bounce b, all frames
    uint32_t peerTarget[kMaxLanes - 1];         // bounce kernel of bounce b: what peer p's done[b - 1][*] add up to once its bounce
                                                // b - 1 of this frame has ended (compared wrap-safe)
    uint32_t* myDone;                // this lane's own counters
    // flushKernel re-arms the count buffer of the frame BEFORE this one for the frame after it, and a peer lane may run one
    // frame behind and still read that buffer: each lane counts its finished frames (flushKernel's last act), and a flush
    // waits (bounded) until every peer has finished the previous frame. (As stream-ordered event waits between the lanes'
    // streams the same dependency cost 3.5 % of a 1080p pass at one sample per tick.)
    uint32_t* myFrameDone;                          // frames this lane has finished, all time
    const uint32_t* peerFrameDone[kMaxLanes - 1];
    uint32_t frameSeq;                              // frames finished before this one (what the peers' counters must have reached)
    uint32_t joinsFrame;                            // 1 in the LAST lane: its flushKernel also waits for the peers' flushes of THIS frame (all of them
                                                    // enqueued before it), so that one event behind it orders the caller's stream after the whole frame
    uint32_t* guardTimeouts;         // incremented when a wait for a peer lane expired (must stay 0). The host reads it at its next
                                     // synchronising call and returns PTSS_ETIMEOUT (ptss_api.hip checkLaneTimeouts)
};

// ---- launchers (ptss_kernels.hip) --------------------------------------------------------------
hipError_t launchRngInit(hipStream_t st, uint32_t* rngHome, uint32_t plane, uint32_t samples, TileMap tile, uint64_t seed,
                         const uint32_t* jumpTable);
hipError_t launchDisplay(hipStream_t st, const FrameBuffers& fb);
hipError_t launchClear(hipStream_t st, const FrameBuffers& fb);
hipError_t launchPrimaryPrep(hipStream_t st, float4* sceneBlob, const SceneLayout& layout, ptss_vec3 origin);
hipError_t launchBounce(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, SceneLayout layout, int bounce,
                        bool isLast, bool sceneInLds, bool bounded, int gridBlocks, TileMap tile, EyeParams eye);
size_t bounceLdsBytes(const SceneLayout& layout, bool sceneInLds);
hipError_t launchFrame(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, SceneLayout layout, int numBounces, bool bounded, int gridBlocks,
                       TileMap tile, EyeParams eye);   // every bounce of a frame in ONE launch (frameKernel): the whole grid must be resident
int frameOccupancyBlocksPerCU(const SceneLayout& layout, bool bounded);
struct FlushTargets {  // flushKernel re-derives the guard of every bounce: target[p][b] = peer p's done total after ITS bounce b of this frame
    uint32_t target[kMaxLanes - 1][kMaxBounces + 1];
};
hipError_t launchFlush(hipStream_t st, const FrameBuffers& fb, int numBounces, const FlushTargets& targets);  // one per lane
int bounceOccupancyBlocksPerCU(const SceneLayout& layout, bool sceneInLds, bool accel);

}  // namespace ptss
