// CudaTracer.h — host half of the reference's CudaTracer.h (constants :3-7, RendererData :13-27,
// ProgramData :32-42, prototypes :44-47). The device half (prototypes :49-89 and every pointer in
// RendererData) lives behind the C-ABI now: ProgramData owns one ptss_context instead of eight raw
// device pointers.
#pragma once
#include <vector>

#include "GPUAnimBitmap.h"
#include "Scene.h"
#include "ptss.h"

#define INVERSE_255 0.00392156862f
#define INVERSE_PI 0.31830988618f
#define RAY_BUMP_EPSILON 1e-4f
#define GAMMA_CORRECTION (1 / 2.2f)

struct RendererData {
    vec3 defaultColor;
    ptss_context* context;  // pointLights/areaLights/rays/spheres/triangles/materials/curandStates of the reference
    size_t numPointLights, numAreaLights, numSpheres, numTriangles;
};

// Multi-GPU (north_star: "frames shard by pixel-tile across the 8 GPUs of one node with an RCCL gather over xGMI of the accumulated
// radiance buffer"): ONE process, one Shard per device — its own context (cfg.device, tileRank / tileWorld = the
// interleaved 8-row bands), stream and display tile — no collective inside the bounce loop, one ncclGather of the uint3
// accumulator tiles to device 0 when a frame is to be looked at (MultiGpu.cpp). The reference is single-GPU; this is its main()
// shape (CudaTracer.cu:649-743) with the device-side allocations repeated per GPU.
struct Shard {
    int device = 0;
    ptss_context* context = nullptr;
    hipStream_t stream = nullptr;
    uchar4* devPixels = nullptr;     // this shard's rows of the display buffer (ptss_alloc_pixels)
    uint32_t* devSend = nullptr;     // the accumulator tile padded to the largest shard (ncclGather wants equal counts)
    void* comm = nullptr;            // ncclComm_t
    size_t localPixels = 0;
    std::vector<int> rows;           // global y of each local row
};

struct ProgramData {
    RendererData renderData;
    std::vector<Shard> shards;       // empty: the single-GPU path of the reference (renderData.context, bitmap.devPixels)
    bool shardsEmulated = false;     // --emulate-gpus: every shard on device 0, the gather by device copies (no RCCL: it refuses two ranks on one device)
    uint32_t* devGather = nullptr;   // device 0: shards x padded tile x 3 words
    size_t paddedPixels = 0;
    int width = 0, height = 0, samplesPerPass = 1;
    Camera camera;
    int lastResetTick;
    int lastTicks = 0;               // the tick of the last generateFrame (the screenshot's divisor)
    unsigned int maxIterations = 15;
    bool resetTicksThisFrame;
    bool usePathTracer = true;
    float lastPassMs = 0.0f;
    bool quiet = false;
};

void generateFrame(uchar4* pixels, void*, int ticks);
void Key(unsigned char key, int x, int y);
bool moveCamera(Camera& camera, unsigned char key);  // HostOps.cpp
void saveScreenshot(char filename[160], int x, int y);
// MultiGpu.cpp
void createShards(ProgramData* data, const ptss_scene_desc& scene, const ptss_render_config& base, int gpus, bool emulateOnOneGpu);
void destroyShards(ProgramData* data);
void generateFrameSharded(ProgramData* data, int ticks);                       // every shard's ptss_generate_frame, then a join
std::vector<uint32_t> gatherAccumulator(ProgramData* data);                    // RCCL gather to device 0 + un-tile: width x height x 3
std::vector<uchar4> displayFromAccumulator(const ProgramData* data, const std::vector<uint32_t>& accum, int ticks);  // CudaTracer.cu:94-98
unsigned long long totalRayBounces(ProgramData* data);

#define PTSS_HANDLE(ans) \
    { ptssAssert((ans), __FILE__, __LINE__); }
inline void ptssAssert(int code, const char* file, int line) {
    if (code != PTSS_OK) {
        fprintf(stderr, "PTSSassert: %s (%s) %s %d\n", ptss_error_string(code), ptss_last_error_detail(), file, line);
        exit(-code);
    }
}
