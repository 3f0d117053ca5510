// GPUAnimBitmap.h — headless stand-in for the reference's display/animation loop
// (CudaTracer/CudaUtils.h:27-188, from "CUDA by Example"). Same struct name, fields and entry points —
// GPUAnimBitmap(w, h, d), anim_and_exit(f, e, k), image_size(), click_drag(f), get_bitmap_ptr(),
// free_resources(), and the static callbacks idle_func / Key / Draw — so the reference's main() keeps
// its shape. What changed, and why:
//   * the GL pixel-buffer object + cudaGraphicsGLRegisterBuffer (CudaUtils.h:72-81) become one
//     hipMalloc'd RGBA buffer: the MI355X box has no display, and the frame callback only ever saw a
//     device pointer anyway (CudaUtils.h:151-154);
//   * glutMainLoop() (never returns) becomes a bounded loop of idle ticks: set_max_ticks(n) before
//     anim_and_exit(); queued key presses (push_key) are delivered between ticks like GLUT would;
//   * errors come back as HIP error codes printed by HIP_ERROR_HANDLE, the twin of CUDA_ERROR_HANDLE
//     (CudaUtils.h:13-21): print + exit(code).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <vector>

#include "ptss_types.h"

// `uchar4` is HIP's own vector type here (as it is CUDA's in the reference); it has the layout of
// ptss_uchar4 and is cast at the C-ABI boundary.
static_assert(sizeof(uchar4) == sizeof(ptss_uchar4), "uchar4 layout");

#ifndef DIM
#define DIM 512  // CudaUtils.h:7
#endif

#define HIP_ERROR_HANDLE(ans) \
    { gpuAssert((ans), __FILE__, __LINE__); }
inline void gpuAssert(hipError_t code, const char* file, int line, bool abort = true) {
    if (code != hipSuccess) {
        fprintf(stderr, "GPUassert: %s %s %d\n", hipGetErrorString(code), file, line);
        if (abort) exit(code);
    }
}

struct GPUAnimBitmap {
    uchar4* devPixels;  // replaces bufferObj/resource (CudaUtils.h:30-31): the mapped-PBO pointer, always mapped
    int width, height;
    void* dataBlock;
    void (*fAnim)(uchar4*, void*, int);
    void (*animExit)(void*);
    void (*clickDrag)(void*, int, int, int, int);
    void (*keyFunc)(unsigned char, int, int);
    int dragStartX, dragStartY;
    int ticks;     // idle_func's `static int ticks = 1` (CudaUtils.h:146), per bitmap here
    int maxTicks;  // headless: how many idle ticks anim_and_exit runs
    std::deque<unsigned char> keys;

    GPUAnimBitmap(int w, int h, void* d) {
        width = w;
        height = h;
        dataBlock = d;
        fAnim = NULL;
        animExit = NULL;
        clickDrag = NULL;
        keyFunc = NULL;
        dragStartX = dragStartY = 0;
        ticks = 1;
        maxTicks = 1;
        devPixels = NULL;
        // cudaChooseDevice + cudaGLSetGLDevice (CudaUtils.h:49-57): first device
        int count = 0;
        HIP_ERROR_HANDLE(hipGetDeviceCount(&count));
        HIP_ERROR_HANDLE(hipSetDevice(0));
        // glBufferData(DIM*DIM*4) + register (CudaUtils.h:72-81)
        HIP_ERROR_HANDLE(hipMalloc((void**)&devPixels, (size_t)image_size()));
        HIP_ERROR_HANDLE(hipMemset(devPixels, 0, (size_t)image_size()));
    }

    ~GPUAnimBitmap() { free_resources(); }

    void free_resources(void) {
        if (devPixels) {
            HIP_ERROR_HANDLE(hipFree(devPixels));
            devPixels = NULL;
        }
    }

    long image_size(void) const { return (long)width * height * 4; }

    void click_drag(void (*f)(void*, int, int, int, int)) { clickDrag = f; }
    void set_max_ticks(int n) { maxTicks = n; }
    void push_key(unsigned char k) { keys.push_back(k); }

    // glutMainLoop stand-in: maxTicks idle callbacks, key presses delivered in between; then the exit hook.
    void anim_and_exit(void (*f)(uchar4*, void*, int), void (*e)(void*), void (*k)(unsigned char key, int x, int y)) {
        GPUAnimBitmap** bitmap = get_bitmap_ptr();
        *bitmap = this;
        fAnim = f;
        animExit = e;
        keyFunc = k;
        for (int i = 0; i < maxTicks; ++i) {
            while (!keys.empty()) {
                const unsigned char key = keys.front();
                keys.pop_front();
                if (keyFunc) keyFunc(key, 0, 0);
            }
            idle_func();
        }
        if (animExit) animExit(dataBlock);
    }

    static GPUAnimBitmap** get_bitmap_ptr(void) {
        static GPUAnimBitmap* gBitmap;
        return &gBitmap;
    }

    // CudaUtils.h:145-159 without the map/unmap pair
    static void idle_func(void) {
        GPUAnimBitmap* bitmap = *(get_bitmap_ptr());
        bitmap->fAnim(bitmap->devPixels, bitmap->dataBlock, bitmap->ticks++);
        Draw();
    }

    // CudaUtils.h:162-171
    static void Key(unsigned char key, int, int) {
        if (key == 27) {
            GPUAnimBitmap* bitmap = *(get_bitmap_ptr());
            if (bitmap->animExit) bitmap->animExit(bitmap->dataBlock);
            bitmap->free_resources();
            exit(0);
        }
    }

    static void Draw(void) {}  // glDrawPixels + glutSwapBuffers: nothing to present without a window

    // what glReadPixels(GL_RGBA) would return: the display buffer, row 0 = bottom
    std::vector<uchar4> read_pixels() const {
        std::vector<uchar4> host((size_t)width * height);
        HIP_ERROR_HANDLE(hipDeviceSynchronize());
        HIP_ERROR_HANDLE(hipMemcpy(host.data(), devPixels, (size_t)image_size(), hipMemcpyDeviceToHost));
        return host;
    }
};
