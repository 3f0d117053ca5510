// MultiGpu.cpp — the host loop of the reference's main() (CudaTracer/CudaTracer.cu:649-743) over N GPUs of one node, in ONE
// process: a context, a stream and a display tile per device (pixel-tile shard: interleaved 8-row bands, RNG bound to the
// global pixel, so the image does not depend on N — include/ptss.h), no collective inside the bounce loop, and ONE RCCL
// gather (ncclGather inside ncclGroupStart / ncclGroupEnd, one call per device of the single-process communicator) of the
// uint3 accumulator tiles to device 0 over xGMI when the frame is to be looked at; un-tiled and tone-scaled on the host
// (CudaTracer.cu:94-98). Calls nothing but the C-ABI (include/ptss.h), the HIP runtime API and RCCL.
//
// Executed so far: N = 1 (communicator, gather and un-tile all run; pixels equal the single-GPU path's), and N = 2 .. 4
// EMULATED on one device (--emulate-gpus: the same shards, streams and un-tile, the gather replaced by device copies because
// RCCL refuses two ranks on one device). N > 1 on N devices: unexecuted (every box of this pipeline has one GPU).
#include <rccl/rccl.h>

#include <chrono>

#include "CudaTracer.h"

#define RCCL_HANDLE(ans)                                                                                   \
    {                                                                                                      \
        ncclResult_t _r = (ans);                                                                           \
        if (_r != ncclSuccess) {                                                                           \
            fprintf(stderr, "RCCLassert: %s %s %d\n", ncclGetErrorString(_r), __FILE__, __LINE__);        \
            exit((int)_r);                                                                                 \
        }                                                                                                  \
    }

void createShards(ProgramData* data, const ptss_scene_desc& scene, const ptss_render_config& base, int gpus, bool emulateOnOneGpu) {
    int count = 0;
    HIP_ERROR_HANDLE(hipGetDeviceCount(&count));
    if (!emulateOnOneGpu && gpus > count) {
        fprintf(stderr, "--gpus %d but %d device(s) visible\n", gpus, count);
        exit(2);
    }
    data->shards.resize((size_t)gpus);
    data->shardsEmulated = emulateOnOneGpu;
    data->width = base.width;
    data->height = base.height;
    data->samplesPerPass = base.samplesPerPass > 0 ? base.samplesPerPass : 1;
    size_t largest = 0;
    for (int k = 0; k < gpus; ++k) {
        Shard& sh = data->shards[(size_t)k];
        sh.device = emulateOnOneGpu ? 0 : k;
        HIP_ERROR_HANDLE(hipSetDevice(sh.device));
        HIP_ERROR_HANDLE(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking));
        ptss_render_config cfg = base;
        cfg.device = sh.device;
        cfg.tileRank = k;
        cfg.tileWorld = gpus;
        cfg.syncEachFrame = 0;   // the shards run side by side; generateFrameSharded joins them
        PTSS_HANDLE(ptss_create(&scene, &cfg, &sh.context));
        PTSS_HANDLE(ptss_set_stream(sh.context, sh.stream));
        PTSS_HANDLE(ptss_local_pixels(sh.context, &sh.localPixels));
        int n = 0;
        PTSS_HANDLE(ptss_local_rows(sh.context, NULL, 0, &n));
        sh.rows.resize((size_t)n);
        PTSS_HANDLE(ptss_local_rows(sh.context, sh.rows.data(), n, &n));
        PTSS_HANDLE(ptss_alloc_pixels(sh.context, reinterpret_cast<ptss_uchar4**>(&sh.devPixels)));
        if (sh.localPixels > largest) largest = sh.localPixels;
    }
    data->paddedPixels = largest;
    for (Shard& sh : data->shards) {
        HIP_ERROR_HANDLE(hipSetDevice(sh.device));
        HIP_ERROR_HANDLE(hipMalloc((void**)&sh.devSend, 3 * largest * sizeof(uint32_t)));
        HIP_ERROR_HANDLE(hipMemset(sh.devSend, 0, 3 * largest * sizeof(uint32_t)));
    }
    HIP_ERROR_HANDLE(hipSetDevice(data->shards[0].device));
    HIP_ERROR_HANDLE(hipMalloc((void**)&data->devGather, (size_t)gpus * 3 * largest * sizeof(uint32_t)));
    if (!emulateOnOneGpu) {   // one communicator object per device of this process
        std::vector<ncclComm_t> comms((size_t)gpus);
        std::vector<int> devs((size_t)gpus);
        for (int k = 0; k < gpus; ++k) devs[(size_t)k] = k;
        RCCL_HANDLE(ncclCommInitAll(comms.data(), gpus, devs.data()));
        for (int k = 0; k < gpus; ++k) data->shards[(size_t)k].comm = comms[(size_t)k];
    }
    data->renderData.context = data->shards[0].context;
}

void destroyShards(ProgramData* data) {
    for (Shard& sh : data->shards) {
        HIP_ERROR_HANDLE(hipSetDevice(sh.device));
        if (sh.comm) RCCL_HANDLE(ncclCommDestroy((ncclComm_t)sh.comm));
        PTSS_HANDLE(ptss_free_pixels(sh.context, reinterpret_cast<ptss_uchar4*>(sh.devPixels)));
        PTSS_HANDLE(ptss_destroy(sh.context));
        HIP_ERROR_HANDLE(hipFree(sh.devSend));
        HIP_ERROR_HANDLE(hipStreamDestroy(sh.stream));
    }
    if (data->devGather) {
        HIP_ERROR_HANDLE(hipSetDevice(data->shards[0].device));
        HIP_ERROR_HANDLE(hipFree(data->devGather));
    }
    data->shards.clear();
}

// One tick on every device (generateFrame's body, CudaTracer.cu:587-647, per shard): the calls only enqueue, so the GPUs run
// side by side; the join at the end is the reference's cudaEventSynchronize (:640) — and the host clock around both is the
// "Time per pass" of the whole frame.
void generateFrameSharded(ProgramData* data, int ticks) {
    const auto t0 = std::chrono::steady_clock::now();
    for (Shard& sh : data->shards) {
        if (data->resetTicksThisFrame) {
            PTSS_HANDLE(ptss_set_camera(sh.context, &data->camera));
            PTSS_HANDLE(ptss_set_mode(sh.context, data->usePathTracer ? 1 : 0));
            PTSS_HANDLE(ptss_set_max_iterations(sh.context, data->maxIterations));
        }
        PTSS_HANDLE(ptss_generate_frame(sh.context, reinterpret_cast<ptss_uchar4*>(sh.devPixels), ticks));
    }
    for (Shard& sh : data->shards) PTSS_HANDLE(ptss_synchronize(sh.context));
    data->lastPassMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// The frame's one collective: every device sends its (padded) accumulator tile, device 0 receives all of them.
std::vector<uint32_t> gatherAccumulator(ProgramData* data) {
    const size_t n = data->shards.size(), words = 3 * data->paddedPixels;
    for (Shard& sh : data->shards) {   // the context's own accumulator -> the padded send buffer, on the shard's stream
        HIP_ERROR_HANDLE(hipSetDevice(sh.device));
        uint32_t* acc = nullptr;
        PTSS_HANDLE(ptss_accumulator_devptr(sh.context, &acc));
        HIP_ERROR_HANDLE(hipMemcpyAsync(sh.devSend, acc, 3 * sh.localPixels * sizeof(uint32_t), hipMemcpyDeviceToDevice, sh.stream));
    }
    if (!data->shardsEmulated) {
        RCCL_HANDLE(ncclGroupStart());
        for (Shard& sh : data->shards)
            RCCL_HANDLE(ncclGather(sh.devSend, data->devGather, words, ncclUint32, 0, (ncclComm_t)sh.comm, sh.stream));
        RCCL_HANDLE(ncclGroupEnd());
    } else {   // rehearsal on one device: what the gather does, as copies
        for (size_t k = 0; k < n; ++k)
            HIP_ERROR_HANDLE(hipMemcpyAsync(data->devGather + k * words, data->shards[k].devSend, words * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                            data->shards[k].stream));
    }
    for (Shard& sh : data->shards) {
        HIP_ERROR_HANDLE(hipSetDevice(sh.device));
        HIP_ERROR_HANDLE(hipStreamSynchronize(sh.stream));
    }
    std::vector<uint32_t> tiles(n * words);
    HIP_ERROR_HANDLE(hipSetDevice(data->shards[0].device));
    HIP_ERROR_HANDLE(hipMemcpy(tiles.data(), data->devGather, tiles.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // un-tile: local row j of shard k is global row rows[j]
    std::vector<uint32_t> full((size_t)3 * data->width * data->height, 0u);
    for (size_t k = 0; k < n; ++k) {
        const Shard& sh = data->shards[k];
        for (size_t j = 0; j < sh.rows.size(); ++j)
            memcpy(&full[(size_t)3 * sh.rows[j] * data->width], &tiles[k * words + 3 * j * data->width], (size_t)3 * data->width * sizeof(uint32_t));
    }
    return full;
}

// the display value of writeToPixelsKernel (CudaTracer.cu:94-98) from the gathered sums: uchar(total * (1.f / samples) + 0.5f), w = 255
std::vector<uchar4> displayFromAccumulator(const ProgramData* data, const std::vector<uint32_t>& accum, int ticks) {
    const float inverseTicks = 1.f / (float)(data->samplesPerPass * (ticks - data->lastResetTick + 1));
    std::vector<uchar4> px((size_t)data->width * data->height);
    for (size_t p = 0; p < px.size(); ++p) {
        px[p].x = (unsigned char)(accum[3 * p + 0] * inverseTicks + 0.5f);
        px[p].y = (unsigned char)(accum[3 * p + 1] * inverseTicks + 0.5f);
        px[p].z = (unsigned char)(accum[3 * p + 2] * inverseTicks + 0.5f);
        px[p].w = 255;
    }
    return px;
}

unsigned long long totalRayBounces(ProgramData* data) {
    if (data->shards.empty()) {
        unsigned long long rays = 0;
        PTSS_HANDLE(ptss_total_ray_bounces(data->renderData.context, &rays));
        return rays;
    }
    unsigned long long sum = 0;
    for (Shard& sh : data->shards) {
        unsigned long long rays = 0;
        PTSS_HANDLE(ptss_total_ray_bounces(sh.context, &rays));
        sum += rays;
    }
    return sum;
}
