/* ptss_host.h — C entry points of libptss_host.so: the HOST-ONLY half of the boundary (no HIP).
 *
 * It exposes the C++ host mirror of the reference (class Scene, CudaTracer/Scene.h:5-27;
 * moveCamera, CudaTracer/CudaTracer.cu:822-870; saveScreenshot, :795-813) to non-C++ callers
 * (the Python test/bench harness), plus read-only probes of the deterministic math and RNG
 * that the device code is built from, so they can be pinned on a machine without a GPU.
 * Every function returns 0 on success and a negative PTSS_HOST_E* code otherwise.
 */
#ifndef PTSS_HOST_H
#define PTSS_HOST_H

#include "ptss_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTSS_HOST_OK 0
#define PTSS_HOST_EINVAL (-1)
#define PTSS_HOST_EIO (-2)

typedef struct ptss_scene ptss_scene; /* owns a C++ Scene */

/* Scene::buildPreset — "default" is Scene::build() (Scene.cpp:17-32); the rest: SURVEY.md §9.6. */
int ptss_scene_create(const char* preset, ptss_scene** out);
void ptss_scene_destroy(ptss_scene* s);
/* Borrowed pointers into the scene's vectors (Scene.h:11-15); valid until destroy. */
int ptss_scene_describe(const ptss_scene* s, ptss_scene_desc* out);

/* Camera() defaults (RenderStructs.h:51-52) and moveCamera (CudaTracer.cu:822-870):
 * key is the reference's key code ('w','a','s','d','q','e','f','h','g','t'); *moved = 1 if handled. */
int ptss_camera_default(ptss_camera* out);
int ptss_camera_move(ptss_camera* cam, unsigned char key, int* moved);

/* saveScreenshot (CudaTracer.cu:795-813): 18-byte header, type 2, 24-bit BGR, bottom-up rows,
 * from a host copy of the RGBA display buffer (row 0 = bottom, as GPUAnimBitmap draws it). */
int ptss_write_tga(const char* filename, const ptss_uchar4* rgba, int width, int height);

/* Pixel-tile ownership used for multi-GPU sharding (north_star): rows are dealt to
 * ranks in bands of band_rows. Returns the number of rows rank owns; fills rows[] (may be NULL). */
int ptss_tile_rows(int height, int band_rows, int rank, int world, int* rows, int cap);

/* Probes (tests only). op: 0 sin, 1 cos, 2 tan, 3 atan, 4 log, 5 exp, 6 pow(x,y), 7 sqrt */
int ptss_probe_math(int op, const float* x, const float* y, float* out, size_t n);
/* 8-bit tone-mapped sample of a radiance value, literal (CudaTracer.cu:72-85), and the 257 thresholds T[0..256] of its
 * table form (csrc/ptquant.h): T[k] = smallest float whose sample is >= k (T[0] = -inf, T[256] = NaN). Returns PTSS_HOST_EINVAL if not monotone. */
int ptss_probe_quantize(const float* x, unsigned int* out, size_t n);
int ptss_probe_quant_table(float* out257);
/* Triangle::intersectRay (Primitives.h:25-83) for n (triangle {v0, e1, e2}, origin, direction, running distance) tuples, in the
 * general form and in the edge-class form the kernels pick for that triangle (csrc/pttri.h); primary != 0: with the
 * camera-origin precomputes of bounce 0. cls[i] = the class; per form six floats: accepted (0/1), dist, b0, b1, b2, det. */
int ptss_probe_triangle_forms(const float* tri9, const float* o3, const float* d3, const float* limit, int primary, size_t n, int* cls,
                              float* general6, float* classed6);
/* XORWOW state after curand_init(seed, subsequence, 0): out6 = v0..v4, d. */
int ptss_probe_rng_init(unsigned long long seed, unsigned int subsequence, unsigned int* out6);
/* n raw draws and the matching (0,1] floats from a state; state advanced in place. */
int ptss_probe_rng_draw(unsigned int* state6, unsigned int* raw, float* uni, size_t n);
/* The 32 subsequence jump matrices A^(2^(67+k)) as 32*160*5 words (images of unit vectors). */
int ptss_probe_rng_jump_table(unsigned int* out, size_t words);

#ifdef __cplusplus
}
#endif
#endif
