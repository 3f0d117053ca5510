// CudaTracer.cpp — generateFrame / Key / saveScreenshot of the reference's app shell
// (CudaTracer/CudaTracer.cu:587-647, :748-786, :795-813) over the C-ABI. All per-frame device work is
// one call, ptss_generate_frame; what stays here is the reference's host-visible behaviour: flags on
// ProgramData, the status line, the key map.
#include "CudaTracer.h"

#include <time.h>

#include <iostream>

#include "HostOps.h"

void generateFrame(uchar4* pixels, void* dataBlock, int ticks) {
    ProgramData* data = (ProgramData*)dataBlock;
    ptss_context* ctx = data->renderData.context;
    data->lastTicks = ticks;
    if (!data->shards.empty()) {   // several GPUs: every shard's generateFrame, side by side (MultiGpu.cpp)
        generateFrameSharded(data, ticks);
        if (data->resetTicksThisFrame) {
            data->lastResetTick = ticks;
            data->resetTicksThisFrame = false;
        }
        if (!data->quiet) {
            std::cout << "Rays per pixel: " << ticks - data->lastResetTick << "  Time per pass: " << data->lastPassMs << "     \r";
            std::cout.flush();
        }
        return;
    }

    // the reference mutates ProgramData from Key(); push those fields down before the frame (:602-608, :620)
    if (data->resetTicksThisFrame) {
        PTSS_HANDLE(ptss_set_camera(ctx, &data->camera));   // also raises the context's reset flag
        PTSS_HANDLE(ptss_set_mode(ctx, data->usePathTracer ? 1 : 0));
        PTSS_HANDLE(ptss_set_max_iterations(ctx, data->maxIterations));
        data->lastResetTick = ticks;
        data->resetTicksThisFrame = false;
    }

    PTSS_HANDLE(ptss_generate_frame(ctx, reinterpret_cast<ptss_uchar4*>(pixels), ticks));   // eye rays, bounce loop, accumulate (:611-642)
    PTSS_HANDLE(ptss_last_pass_ms(ctx, &data->lastPassMs));

    if (!data->quiet) {  // :645-646
        std::cout << "Rays per pixel: " << ticks - data->lastResetTick << "  Time per pass: " << data->lastPassMs << "     \r";
        std::cout.flush();
    }
}

void Key(unsigned char key, int, int) {
    GPUAnimBitmap* bitmap = *(GPUAnimBitmap::get_bitmap_ptr());
    ProgramData* data = (ProgramData*)bitmap->dataBlock;
    switch (key) {
        case 27:  // Esc (:753-759)
            if (bitmap->animExit) bitmap->animExit(bitmap->dataBlock);
            bitmap->free_resources();
            exit(0);
        case 32:  // space: path tracer <-> ray tracer (:760-765) ...
            data->usePathTracer = !data->usePathTracer;
            data->resetTicksThisFrame = true;
            // ... and, like the reference, no break: falls into the screenshot case
        case 48: {  // '0': screenshot named after the time (:766-779)
            time_t now = time(0);
            struct tm tstruct = *localtime(&now);
            char buf[160];
            strftime(buf, sizeof(buf), "renders/render%Y-%m-%d-%H%M%S.tga", &tstruct);
            saveScreenshot(buf, bitmap->width, bitmap->height);
        }
    }
    if (moveCamera(data->camera, key)) data->resetTicksThisFrame = true;  // :782-785
}

// :795-813 — glReadPixels becomes a device-to-host copy of the display buffer
void saveScreenshot(char filename[160], int x, int y) {
    GPUAnimBitmap* bitmap = *(GPUAnimBitmap::get_bitmap_ptr());
    if (!bitmap || bitmap->width != x || bitmap->height != y) return;
    ProgramData* data = (ProgramData*)bitmap->dataBlock;
    // several GPUs: the frame is on N devices — one RCCL gather of the accumulator tiles, un-tiled and scaled on the host
    const std::vector<uchar4> host = data->shards.empty() ? bitmap->read_pixels() : displayFromAccumulator(data, gatherAccumulator(data), data->lastTicks);
    if (!writeTga(filename, reinterpret_cast<const ptss_uchar4*>(host.data()), x, y)) fprintf(stderr, "saveScreenshot: cannot write %s\n", filename);
}
