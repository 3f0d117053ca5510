/* ptss.h — C-ABI of libptss.so: the MI355X-native drop-in for the reference's per-frame hot path.
 *
 * The reference has no plugin/FFI layer; its seam is the GPUAnimBitmap frame callback plus the
 * scene vectors. Each entry point below names the reference code it replaces
 * (paths relative to /root/reference/CudaTracer/). INTEGRATION.md shows the host-side glue.
 *
 * Conventions: plain pointers and sizes only; every function returns PTSS_OK (0) or a negative
 * PTSS_E* code (no exit(), unlike CUDA_ERROR_HANDLE, CudaUtils.h:13-21); no exceptions cross the
 * boundary; one host thread per context at a time (the reference is single-threaded, CudaUtils.h:117).
 * There is NO CPU fallback: without a usable HIP device ptss_create fails with PTSS_ENODEVICE.
 */
#ifndef PTSS_H
#define PTSS_H

#include "ptss_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTSS_OK 0
#define PTSS_EINVAL (-1)   /* bad argument */
#define PTSS_EHIP (-2)     /* a HIP runtime call or kernel launch failed; see ptss_last_error_detail */
#define PTSS_ENODEVICE (-3)/* no usable HIP device */
#define PTSS_ENOMEM (-4)   /* host or device allocation failed */
#define PTSS_ERANGE (-5)   /* output buffer too small / index out of range */
#define PTSS_ETIMEOUT (-6) /* a bounded wait on the device (~2 s) expired: a frame lane for a peer lane, or a workgroup of the one-launch
                              frame kernel for its shard; the frame was not traced as specified, the buffers may hold another image */
#define PTSS_VERSION 300   /* what ptss_version() of a matching library returns; bumped whenever a struct below changes */

typedef struct ptss_context ptss_context; /* ≙ ProgramData + RendererData + every cudaMalloc of main() */

typedef struct ptss_render_config {
    unsigned int structSize;     /* sizeof(ptss_render_config) as the CALLER compiled it (ptss_default_config fills it in);
                                    ptss_create refuses any other value, so a binding built against an older header fails
                                    loudly instead of being read past its end */
    int width, height;           /* full frame; the reference's compile-time DIM x DIM (CudaUtils.h:7) */
    unsigned long long seed;     /* curand_init seed; the reference passes clock64() (CudaTracer.cu:28) */
    unsigned int maxIterations;  /* ProgramData::maxIterations, default 15 (CudaTracer.h:39) */
    int device;                  /* HIP device ordinal (reference: cudaChooseDevice, CudaUtils.h:49-57) */
    /* Pixel-tile shard (north_star): this context owns the rows y with
     * (y / bandRows) % tileWorld == tileRank. tileWorld = 1 renders the whole frame. The RNG
     * subsequence is the GLOBAL pixel index, so the image does not depend on the sharding — with ONE exception:
     * the reference stops bouncing once <= 128 rays are alive in the WHOLE frame (CudaTracer.cu:622); a shard cannot
     * know that count without a collective per bounce, so a context with tileWorld > 1 never stops early (and
     * ptss_live_counts reports its own rays down to 0). Sharded and unsharded images are identical whenever more than
     * 128 rays stay alive frame-wide at every bounce that runs (always, at the benchmark sizes); they differ in tiny
     * frames and in frames whose last few rays outlive the guard (tests/test_gpu_tiles.py pins both behaviours). */
    int tileRank, tileWorld, bandRows;
    int syncEachFrame;     /* 1: block on the stop event and record ms each frame, as CudaTracer.cu:639-642 */
    int floatAccumulator;  /* 1: also keep a linear float32 sum of radiance0 per pixel */
    int timeKernels;       /* 1: bracket every bounce kernel with HIP events (bench.py roofline) */
    /* Extension, default 1 = the reference: S independent samples per pixel per
     * ptss_generate_frame call. Sample lane l of global pixel g owns XORWOW subsequence g*S + l; every sample
     * is still tone-mapped on its own before it is summed (CudaTracer.cu:72-92); the display divides by
     * S*(ticks - lastResetTick + 1). Lets one launch carry S times the rays (multi-GPU shards stay busy). 1..64. */
    int samplesPerPass;
    /* Scenes with >= 64 finite spheres are traversed through spatially sorted sphere chunks (same image
     * as the reference's every-sphere loop, tests/test_gpu_many_spheres.py). 1 keeps the every-sphere loop for them too. */
    int everySphereLoop;
    /* Frame lanes: the frame traced as K ray populations on K streams of the device, the tail of one lane's launches
     * overlapping the other lanes' kernels. The image — loop guard included — does not depend on K.
     * 0 = the library's choice: ONE lane, unless lanesFreeRun is set (then 2 for 3*2^17..2^24 rays per pass, e.g. 800x600 ...
     * 3840x2160 at one sample per tick, else 1); 1..4 = that many. */
    int frameLanes;
    /* Stream ordering of a context with more than one lane. 0 (default): STRICT — like a one-lane context the call is
     * ordered on the caller's stream: the lanes of every frame start behind whatever the caller enqueued on its stream before
     * the call, and the stream continues behind the whole frame, so work enqueued between two ptss_generate_frame calls
     * (a copy of dev_pixels, a gather or a zero-fill of the bound accumulator, ...) sees exactly the frames before it. With that
     * fork and join every frame two lanes are no faster than one (measured, profiles/README.md). 1: FREE-RUNNING — the lanes are
     * forked from the caller's stream only on a reset or camera change and run on from frame to frame (this is where lanes
     * gain: +15-18 % at 1080p, one sample per tick); the caller's stream still continues behind each frame, but the NEXT frame's
     * kernels do not wait for anything the caller enqueued after the previous call. Opt in only if, between two calls,
     * nothing touches dev_pixels, the accumulator or the float sums on the device — or call ptss_synchronize() /
     * ptss_request_reset() first. */
    int lanesFreeRun;
    /* One launch per frame, opt-in: a frame small enough for ALL its bounce-0 tiles to be resident on the device
     * at once (up to about 3*2^17 rays per pass on an MI355X: 512x512 or 640x480 at one sample per tick) is traced by ONE kernel
     * whose workgroups carry their shard from bounce to bounce, instead of one launch per bounce each at most one resident
     * round wide: +7 % at 512x512. Same image, same counters. 1 = on where the frame qualifies (one lane, scene in LDS; a frame
     * that does not qualify is traced bounce by bounce all the same); 0 (default) / -1 = off. Opt-in because the kernel's
     * workgroups wait for each other: it must have the device to itself — two such kernels running at once (two contexts on
     * two streams, or two processes sharing the GPU) can each hold slots the other needs; every wait is bounded (~2 s) and an
     * expired one surfaces as PTSS_ETIMEOUT, but the frame is then lost. */
    int oneLaunchFrames;
} ptss_render_config;

/* Fills the reference's defaults: 512x512 (DIM), maxIterations 15, seed 0x5EED, one tile, sync on. */
int ptss_default_config(ptss_render_config* cfg);

/* ≙ main(): cudaMalloc x8, cudaMemcpy x5 of the scene vectors, ProgramData fill, curandSetupKernel
 * (CudaTracer.cu:671-724). Scene arrays are copied; the caller may free them afterwards.
 * Starts with Camera() defaults, usePathTracer = true, resetTicksThisFrame = true (:703,:717). */
int ptss_create(const ptss_scene_desc* scene, const ptss_render_config* cfg, ptss_context** out);

/* ≙ cudaFree x8 + delete data (CudaTracer.cu:731-740). */
int ptss_destroy(ptss_context* ctx);

/* ≙ generateFrame(uchar4* pixels, void* dataBlock, int ticks) (CudaTracer.cu:587-647), the fAnim
 * callback of GPUAnimBitmap (CudaUtils.h:36,154). dev_pixels: DEVICE pointer to this context's
 * local pixels (width * ptss_local_rows, RGBA, row 0 = bottom), owned by the caller, or NULL to
 * skip the display write. ticks: the caller's running counter (starts at 1, CudaUtils.h:146). */
int ptss_generate_frame(ptss_context* ctx, ptss_uchar4* dev_pixels, int ticks);

/* ≙ Key()/moveCamera() effects on ProgramData (CudaTracer.cu:782-785): store camera, set the reset flag. */
int ptss_set_camera(ptss_context* ctx, const ptss_camera* camera);
int ptss_get_camera(const ptss_context* ctx, ptss_camera* out);
/* ≙ resetTicksThisFrame = true (CudaTracer.cu:764,784). */
int ptss_request_reset(ptss_context* ctx);
/* ≙ space bar: usePathTracer toggle + reset (CudaTracer.cu:760-765). 0 = one-bounce ray tracing. */
int ptss_set_mode(ptss_context* ctx, int usePathTracer);
int ptss_set_max_iterations(ptss_context* ctx, unsigned int maxIterations);

/* Plumbing for callers that own device memory / streams (PyTorch, GPUAnimBitmap). */
int ptss_set_stream(ptss_context* ctx, void* hipStream);            /* default: the null stream */
int ptss_bind_accumulator(ptss_context* ctx, uint32_t* dev_uint3);  /* external totalPixelColors, 3 u32 per local pixel */
int ptss_accumulator_devptr(ptss_context* ctx, uint32_t** out);
int ptss_float_accumulator_devptr(ptss_context* ctx, float** out);
int ptss_alloc_pixels(ptss_context* ctx, ptss_uchar4** out_dev);    /* headless stand-in for the GL PBO (CudaUtils.h:72-81) */
int ptss_free_pixels(ptss_context* ctx, ptss_uchar4* dev);
int ptss_local_pixels(const ptss_context* ctx, size_t* out);        /* width * local rows */
int ptss_local_rows(const ptss_context* ctx, int* rows, int cap, int* count); /* global y of each local row */

/* Device -> host copies (synchronising). count = number of ELEMENTS of the destination type. */
int ptss_read_accumulator(ptss_context* ctx, uint32_t* host_uint3, size_t count);       /* 3 * local pixels */
int ptss_read_float_accumulator(ptss_context* ctx, float* host_float3, size_t count);   /* 3 * local pixels */
int ptss_read_pixels(ptss_context* ctx, const ptss_uchar4* dev_pixels, ptss_uchar4* host, size_t count);
int ptss_read_rng_state(ptss_context* ctx, size_t local_pixel, uint32_t* out6);         /* v0..v4, d (sample lane 0) */
int ptss_read_rng_state_lane(ptss_context* ctx, size_t local_pixel, unsigned int lane, uint32_t* out6);
int ptss_synchronize(ptss_context* ctx);

/* ≙ the "Rays per pixel / Time per pass" line (CudaTracer.cu:641-646). */
int ptss_last_pass_ms(ptss_context* ctx, float* out_ms);
int ptss_samples_since_reset(const ptss_context* ctx, int* out);
/* Rays that ENTERED each bounce of the last frame (numRays at CudaTracer.cu:623); returns count in *n. */
int ptss_live_counts(ptss_context* ctx, uint32_t* out, int cap, int* n);
/* Sum over all frames and bounces of rays processed by the bounce kernel (the Mrays numerator). */
int ptss_total_ray_bounces(ptss_context* ctx, unsigned long long* out);
/* HIP-event time of the bounce kernel since the last call (needs cfg.timeKernels): total ms, launches. */
int ptss_bounce_kernel_time(ptss_context* ctx, double* total_ms, unsigned long long* launches);

/* 1 when this context traces a frame with ONE launch (cfg.oneLaunchFrames resolved for the scene image in use). */
int ptss_one_launch_frames(const ptss_context* ctx, int* out);
/* Frame lanes this context runs (cfg.frameLanes resolved). */
int ptss_frame_lanes(const ptss_context* ctx, int* out);
/* How often a lane gave up waiting for a peer lane's live count (a stream that did not run for ~2 s): always 0 in a healthy
 * process; a non-zero value means the loop guard of that frame was decided without the peer. Every synchronising entry
 * point (ptss_synchronize, ptss_read_*, ptss_live_counts, ptss_total_ray_bounces, ptss_bounce_kernel_time, and
 * ptss_generate_frame with syncEachFrame) returns PTSS_ETIMEOUT once when this count has grown since the last check. */
int ptss_guard_timeouts(ptss_context* ctx, unsigned int* out);

/* Diagnostic builds only (-DPTSS_DIAG=<bits>, csrc/ptss_diag.h, tools/build_variants.py): the eight counter words of that
 * build (sphere candidates per lane, scatter blocks, chunk culling, shadow-segment pairs, queue lengths); all zero in the
 * shipped library, which carries no counter. */
int ptss_debug_counters(ptss_context* ctx, unsigned long long* out8);

const char* ptss_error_string(int code);
const char* ptss_last_error_detail(void);
int ptss_version(void);

#ifdef __cplusplus
}
#endif
#endif
