// main.cpp — the reference's main() (CudaTracer/CudaTracer.cu:649-743) on the MI355X drop-in.
// Same sequence: build the Scene, make ProgramData + GPUAnimBitmap, hand the scene vectors to the
// device (one ptss_create instead of cudaMalloc x8 + cudaMemcpy x5 + curandSetupKernel), then
// bitmap.anim_and_exit(generateFrame, NULL, Key). The reference ignores argv; this build accepts
// optional overrides with the reference's values as defaults:
//   ptss_main [--preset default] [--size 512x512] [--ticks 16] [--bounces 15] [--seed N] [--samples-per-pass S]
//             [--keys "wwd f"] [--out image.tga] [--quiet]
//             [--gpus N]            the frame sharded by pixel tile over N GPUs of this node, one RCCL gather (MultiGpu.cpp)
//             [--emulate-gpus N]    the same N shards on device 0, the gather as device copies (rehearsal on a one-GPU box)
#include <stdlib.h>
#include <string.h>

#include <string>

#include "CudaTracer.h"
#include "HostOps.h"

int main(int argc, char* argv[]) {
    std::string preset = "default", out, keys;
    int width = DIM, height = DIM, ticks = 16, gpus = 0, samples = 1;
    bool emulate = false;
    unsigned bounces = 15;
    unsigned long long seed = 0x5EED;
    bool quiet = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--preset") preset = next();
        else if (a == "--size") { if (sscanf(next(), "%dx%d", &width, &height) != 2) { fprintf(stderr, "bad --size\n"); return 2; } }
        else if (a == "--ticks") ticks = atoi(next());
        else if (a == "--bounces") bounces = (unsigned)atoi(next());
        else if (a == "--seed") seed = strtoull(next(), NULL, 0);
        else if (a == "--keys") keys = next();
        else if (a == "--out") out = next();
        else if (a == "--quiet") quiet = true;
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--emulate-gpus") { gpus = atoi(next()); emulate = true; }
        else if (a == "--samples-per-pass") samples = atoi(next());
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }

    Scene scene;
    vec3 defaultColor = v3(0, 0, 0);
    if (!scene.buildPreset(preset)) {  // scene.build() for "default"
        fprintf(stderr, "unknown preset %s\n", preset.c_str());
        return 2;
    }

    // initialize bitmap and data
    ProgramData* data = new ProgramData();
    GPUAnimBitmap bitmap(width, height, data);

    // allocate GPU memory, copy the scene, seed the per-pixel random streams (CudaTracer.cu:671-724)
    ptss_render_config cfg;
    PTSS_HANDLE(ptss_default_config(&cfg));
    cfg.width = width;
    cfg.height = height;
    cfg.maxIterations = bounces;
    cfg.seed = seed;
    cfg.samplesPerPass = samples;
    const ptss_scene_desc desc = scene.desc(defaultColor);
    ptss_context* ctx = NULL;
    if (gpus > 0) {   // one context, stream and display tile per GPU; RCCL communicator over them
        createShards(data, desc, cfg, gpus, emulate);
        ctx = data->renderData.context;
    } else {
        PTSS_HANDLE(ptss_create(&desc, &cfg, &ctx));
    }

    // put values in a data block (:703-717)
    data->camera = Camera();
    data->renderData.context = ctx;
    data->renderData.numPointLights = scene.pointLightsVec.size();
    data->renderData.numAreaLights = scene.areaLightsVec.size();
    data->renderData.numSpheres = scene.spheresVec.size();
    data->renderData.numTriangles = scene.trianglesVec.size();
    data->renderData.defaultColor = defaultColor;
    data->maxIterations = bounces;
    data->resetTicksThisFrame = true;
    data->quiet = quiet;

    bitmap.set_max_ticks(ticks);
    for (char k : keys) bitmap.push_key((unsigned char)k);
    bitmap.anim_and_exit((void (*)(uchar4*, void*, int))generateFrame, NULL, (void (*)(unsigned char, int, int))Key);

    if (!quiet) printf("\n");
    if (!out.empty()) {
        char name[160];
        strncpy(name, out.c_str(), sizeof(name) - 1);
        name[sizeof(name) - 1] = 0;
        saveScreenshot(name, width, height);
    }
    const unsigned long long rays = totalRayBounces(data);
    printf("%d ticks, %llu ray-bounces, last pass %.3f ms", ticks, rays, data->lastPassMs);
    if (gpus > 0) printf(", %d shard(s) on %s, gathered by %s", gpus, emulate ? "device 0" : "as many GPUs", emulate ? "device copies" : "ncclGather");
    printf("\n");

    // free (:731-740)
    if (gpus > 0) destroyShards(data);
    else PTSS_HANDLE(ptss_destroy(ctx));
    bitmap.free_resources();
    delete data;
    return 0;
}
