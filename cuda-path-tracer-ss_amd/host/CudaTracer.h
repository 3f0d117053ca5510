// CudaTracer.h — host half of the reference's CudaTracer.h (constants :3-7, RendererData :13-27,
// ProgramData :32-42, prototypes :44-47). The device half (prototypes :49-89 and every pointer in
// RendererData) lives behind the C-ABI now: ProgramData owns one ptss_context instead of eight raw
// device pointers.
#pragma once
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

struct ProgramData {
    RendererData renderData;
    Camera camera;
    int lastResetTick;
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

#define PTSS_HANDLE(ans) \
    { ptssAssert((ans), __FILE__, __LINE__); }
inline void ptssAssert(int code, const char* file, int line) {
    if (code != PTSS_OK) {
        fprintf(stderr, "PTSSassert: %s (%s) %s %d\n", ptss_error_string(code), ptss_last_error_detail(), file, line);
        exit(-code);
    }
}
