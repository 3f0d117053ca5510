// oracle.cpp — CPU restatement of the reference's per-frame wavefront path tracer.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT. *** Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library. The product (libptss.so) never calls it and has no CPU
// fallback.
//
// PARITY STATUS: "parity unpinned by the reference" — the reference ships no tests, golden vectors
// or fixtures for this path, it cannot be compiled here (needs nvcc, cuRAND,
// Thrust, glm, GLUT/GLEW: SURVEY.md §8c), and even on CUDA it is not reproducible (clock64() seed,
// unstable partition, uninitialised roughness). What pins this oracle instead: analytic known-answer
// tests (tests/test_oracle_kat.py), rocRAND's independent XORWOW tables (tests/test_xorwow.py), libm
// bounds on the shared math (tests/test_ptmath.py) and committed golden vectors (tests/golden/).
//
// It follows, function by function (all paths relative to /root/reference/CudaTracer/):
//   generateFrame                       CudaTracer.cu:587-647
//   curandSetupKernel / curand_init     CudaTracer.cu:22-29
//   clearPixels                         CudaTracer.cu:31-49
//   computeEyeRaysKernel/computeEyeRay  CudaTracer.cu:51-61, 321-343
//   pathTraceKernel                     CudaTracer.cu:106-206
//   computeIndirectRadianceAndScatter   CudaTracer.cu:208-318
//   shade / getAreaLightPoint / lineOfSight   CudaTracer.cu:345-390, 392-418, 420-455
//   Fresnel, Snell, reflRay x2, refrRay CudaTracer.cu:457-531
//   randomDirection{Lambert,Phong,Beckmann}, rotateVectorToVector   CudaTracer.cu:533-585
//   writeToPixelsKernel                 CudaTracer.cu:63-104
//   Triangle::intersectRay              Primitives.h:25-83
//   Sphere::intersectRay/getSurfaceElement   Primitives.h:98-175
//   thrust::partition + ray_is_active   CudaTracer.cu:629, CudaTracer.h:91-98 (here: STABLE partition)
// with the pinned decisions of SURVEY.md §9 (RNG bound to the pixel; all live rays processed;
// roughness defaults to 0; non-square generalisation of the eye ray).
//
// Arithmetic: IEEE f32, no contraction; dot/cross use the fma chains of ptmath.h; transcendentals
// are ptm::* (shared with the kernels by design, see ptmath.h header). Data layout is the
// reference's AoS `Ray` + one RNG state per pixel — deliberately unlike the device code's SoA pools.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#include <algorithm>
#include <vector>

#ifdef ORACLE_LIBM_MATH   // second build, sharing no arithmetic with the product: oracle/libm_math.h
#include "libm_math.h"
#else
#include "ptmath.h"
#endif
#include "ptss_types.h"

using namespace ptv;

namespace {

// ---- RNG: own statement of XORWOW (does not include csrc/xorwow.h) -------------------------------
struct CurandState {
    uint32_t d;
    uint32_t v[5];
};

inline uint32_t curand(CurandState& s) {
    const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1];
    s.v[1] = s.v[2];
    s.v[2] = s.v[3];
    s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}

// (0,1]: x * 2^-32 + 2^-33
inline float curand_uniform(CurandState* s) {
    return (float)curand(*s) * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

// 160x160 GF(2) matrix as 160 column images.
struct BitMat {
    uint32_t col[160][5];
};

void matVec(const BitMat& m, uint32_t v[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 160; ++i)
        if ((v[i >> 5] >> (i & 31)) & 1u)
            for (int w = 0; w < 5; ++w) r[w] ^= m.col[i][w];
    memcpy(v, r, sizeof(r));
}

void matSquare(const BitMat& m, BitMat& out) {
    for (int i = 0; i < 160; ++i) {
        memcpy(out.col[i], m.col[i], sizeof(out.col[i]));
        matVec(m, out.col[i]);
    }
}

// powers[k] = A^(2^(67+k)): one subsequence = 2^67 draws (curand_init's `sequence` argument).
const std::vector<BitMat>& sequencePowers() {
    static std::vector<BitMat> powers;
    if (!powers.empty()) return powers;
    BitMat a, b;
    for (int i = 0; i < 160; ++i) {
        CurandState e;
        memset(&e, 0, sizeof(e));
        e.v[i >> 5] = 1u << (i & 31);
        (void)curand(e);
        memcpy(a.col[i], e.v, sizeof(e.v));
    }
    for (int s = 0; s < 67; ++s) {
        matSquare(a, b);
        a = b;
    }
    for (int k = 0; k < 32; ++k) {
        powers.push_back(a);
        matSquare(a, b);
        a = b;
    }
    return powers;
}

// curand_init(seed, sequence, 0, &state)
void curand_init(uint64_t seed, uint32_t sequence, CurandState* st) {
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    st->d = 6615241u + t1 + t0;
    st->v[0] = 123456789u + t0;
    st->v[1] = 362436069u ^ t0;
    st->v[2] = 521288629u + t1;
    st->v[3] = 88675123u ^ t1;
    st->v[4] = 5783321u + t0;
    const std::vector<BitMat>& p = sequencePowers();
    for (int k = 0; k < 32; ++k)
        if ((sequence >> k) & 1u) matVec(p[k], st->v);
}

// ---- reference records -------------------------------------------------------------------------
struct Ray {  // RenderStructs.h:24-39
    vec3 origin;
    vec3 direction;
    vec3 radiance0;
    vec3 radiance1;
    int pixelOffset;
    bool active;
    int lane;  // extension: sample lane 0..S-1 (S = samplesPerPass; the reference has S = 1)
};

inline Ray makeRay(vec3 origin, vec3 direction) {
    Ray r;
    r.origin = origin;
    r.direction = direction;
    r.radiance0 = v3(0, 0, 0);
    r.radiance1 = v3(1, 1, 1);
    r.pixelOffset = 0;
    r.active = true;
    r.lane = 0;
    return r;
}

struct SurfaceElement {  // RenderStructs.h:110-121
    vec3 point;
    vec3 normal;
    int materialIdx;
};

struct RendererData {  // CudaTracer.h:13-27
    vec3 defaultColor;
    std::vector<ptss_point_light> pointLights;
    std::vector<ptss_area_light> areaLights;
    std::vector<ptss_sphere> spheres;
    std::vector<ptss_triangle> triangles;
    std::vector<ptss_material> materials;
};

const float RAY_BUMP_EPSILON = 1e-4f;       // CudaTracer.h:6
const float GAMMA_CORRECTION = (1 / 2.2f);  // CudaTracer.h:7
const float INVERSE_PI = 0.31830988618f;    // CudaTracer.h:4
const float M_PI_F = 3.14159265358979323846f;  // RenderStructs.h:9

// ---- Primitives.h:25-83 -------------------------------------------------------------------------
bool triangleIntersectRay(const ptss_triangle& tri, const Ray& ray, float& distance, SurfaceElement& surfel,
                          bool updateSurfel = true) {
    const float epsilon = 1e-7f;
    float weight[3];

    const vec3 e1 = tri.vertex1 - tri.vertex0;
    const vec3 e2 = tri.vertex2 - tri.vertex0;
    const vec3 q = cross(ray.direction, e2);
    const float det = dot(e1, q);
    if (ptm::abs(det) <= epsilon) return false;

    const float inverseDet = 1 / det;
    const vec3 s = ray.origin - tri.vertex0;
    const vec3 r = cross(s, e1);
    const float dist = dot(e2, r) * inverseDet;
    if ((dist <= 0.0f) || (dist > distance)) return false;

    weight[1] = dot(s, q) * inverseDet;
    weight[2] = dot(ray.direction, r) * inverseDet;
    weight[0] = 1.0f - (weight[1] + weight[2]);
    if ((weight[0] < 0) || (weight[1] < 0) || (weight[2] < 0)) return false;

    if (updateSurfel) {
        const vec3 normalResult = (tri.normal0 * weight[0] + tri.normal1 * weight[1]) + tri.normal2 * weight[2];
        const vec3 intersectionPoint = ray.origin + ray.direction * dist;
        surfel.point = intersectionPoint;
        surfel.normal = normalResult;
        surfel.materialIdx = tri.materialIdx;
    }
    distance = dist;
    return true;
}

// ---- Primitives.h:98-175 ------------------------------------------------------------------------
SurfaceElement sphereSurfaceElement(const ptss_sphere& sp, const Ray& ray, float distance) {
    SurfaceElement se;
    se.point = ray.origin + ray.direction * distance;
    se.normal = normalize(se.point - sp.position);
    se.materialIdx = sp.materialIdx;
    return se;
}

bool sphereIntersectRay(const ptss_sphere& sp, const Ray& ray, float& distance, SurfaceElement& surfel,
                        bool updateSurfel = true) {
    const vec3 v = ray.origin - sp.position;
    const float b = dot(ray.direction, v) * 2;
    const float c = dot(v, v) - (sp.radius * sp.radius);
    float discriminent = (b * b) - 4 * c;
    if (discriminent < 0) return false;

    discriminent = ptm::sqrt(discriminent);
    float t0 = (-b + discriminent) * 0.5f;
    float t1 = (-b - discriminent) * 0.5f;
    if (t0 < 0 && t1 < 0) return false;
    if (t0 > t1) {
        const float temp = t0;
        t0 = t1;
        t1 = temp;
    }
    if (t0 < 0) {  // origin inside the sphere
        if (t1 > distance) return false;
        distance = t1;
    } else {
        if (t0 > distance) return false;
        distance = t0;
    }
    if (updateSurfel) surfel = sphereSurfaceElement(sp, ray, distance);
    return true;
}

// ---- CudaTracer.cu:579-585 -----------------------------------------------------------------------
quat rotateVectorToVector(const vec3& source, const vec3& target) {
    const vec3 axis = cross(source, target);
    return normalize(q4(1.0f + dot(source, target), axis.x, axis.y, axis.z));
}

// ---- CudaTracer.cu:533-545 -----------------------------------------------------------------------
vec3 randomDirectionLambert(const vec3& normal, CurandState& state) {
    const float theta = curand_uniform(&state) * 2 * M_PI_F;
    const float s = curand_uniform(&state);
    const float y = ptm::sqrt(s);
    const float r = ptm::sqrt(1 - y * y);
    float sn, cs;
    ptm::sincos(theta, sn, cs);
    const vec3 sample = v3(r * cs, y, r * sn);
    return rotate(rotateVectorToVector(v3(0, 1, 0), normal), sample);
}

// ---- CudaTracer.cu:547-559 -----------------------------------------------------------------------
vec3 randomDirectionPhong(const vec3& w_o, float exponent, CurandState& state) {
    const float theta = curand_uniform(&state) * 2 * M_PI_F;
    const float s = curand_uniform(&state);
    const float y = ptm::pow(s, 1 / (exponent + 1));
    const float r = ptm::sqrt(1 - y * y);
    float sn, cs;
    ptm::sincos(theta, sn, cs);
    const vec3 sample = v3(r * cs, y, r * sn);
    return rotate(rotateVectorToVector(v3(0, 1, 0), w_o), sample);
}

// ---- CudaTracer.cu:561-577 -----------------------------------------------------------------------
vec3 randomDirectionBeckmann(const vec3& normal, float roughness, CurandState& state) {
    const float theta = ptm::atan(-roughness * roughness * ptm::log(1.0f - curand_uniform(&state)));
    const float phi = curand_uniform(&state) * 2 * M_PI_F;
    float sinPhi, cosPhi, sinTheta, cosTheta;
    ptm::sincos(phi, sinPhi, cosPhi);
    ptm::sincos(theta, sinTheta, cosTheta);
    const vec3 m = v3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
    return rotate(rotateVectorToVector(v3(0, 1, 0), normal), m);
}

// ---- CudaTracer.cu:457-472 -----------------------------------------------------------------------
float computeFresnelForReflectance(float cosI, float sinT2, float n1, float n2, float /*n*/) {
    if (sinT2 > 1.0f) return 1.0f;
    const float cosT = ptm::sqrt(1.0f - sinT2);
    const float r_s = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    const float r_p = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
    return (r_s * r_s + r_p * r_p) * 0.5f;
}

// ---- CudaTracer.cu:474-494 (cosI is flipped IN PLACE when the ray is inside) ---------------------
void computeSinT2AndRefractiveIndexes(float refrIndex, float& cosI, float& sinT2, float& n1, float& n2, float& n) {
    if (cosI > 0) {
        n2 = refrIndex;
        n1 = 1.0f;
    } else {
        cosI = -cosI;
        n1 = refrIndex;
        n2 = 1.0f;
    }
    n = n1 / n2;
    sinT2 = n * n * (1.0f - cosI * cosI);
}

// ---- CudaTracer.cu:496-503 -----------------------------------------------------------------------
void reflRay(Ray& ray, const SurfaceElement& surfel, float cosI) {
    const vec3 w_o = ray.direction - (2 * (-cosI)) * surfel.normal;
    ray.origin = surfel.point + (surfel.normal * RAY_BUMP_EPSILON);
    ray.direction = w_o;
}

// ---- CudaTracer.cu:505-514 -----------------------------------------------------------------------
void reflRay(Ray& ray, const vec3& point, const vec3& normal) {
    const float cosI = ptm::abs(dot(ray.direction, normal));
    const vec3 w_o = ray.direction - (2 * (-cosI)) * normal;
    ray.origin = point + (normal * RAY_BUMP_EPSILON);
    ray.direction = w_o;
}

// ---- CudaTracer.cu:516-531 -----------------------------------------------------------------------
void refrRay(Ray& ray, const SurfaceElement& surfel, float cosI, float sinT2, float n) {
    if (sinT2 > 1.0f) ray.active = false;
    const float cosT = ptm::sqrt(1.0f - sinT2);
    const vec3 w_o = normalize(n * ray.direction + (n * cosI - cosT) * surfel.normal);
    ray.origin = surfel.point + (w_o * RAY_BUMP_EPSILON);
    ray.direction = w_o;
}

// ---- CudaTracer.cu:420-455 -----------------------------------------------------------------------
bool lineOfSight(const RendererData& data, const vec3& normal, const vec3& point0, const vec3& point1, vec3& w_i,
                 float& distance2) {
    const vec3 offset = point1 - point0;
    distance2 = dot(offset, offset);
    float distance = ptm::sqrt(distance2);
    w_i = offset / distance;
    const Ray losRay = makeRay(point0 + (RAY_BUMP_EPSILON * normal), w_i);
    distance -= 2 * RAY_BUMP_EPSILON;

    SurfaceElement surfel;
    for (size_t i = 0; i < data.spheres.size(); i++)
        if (sphereIntersectRay(data.spheres[i], losRay, distance, surfel, false)) return false;
    for (size_t i = 0; i < data.triangles.size(); i++)
        if (triangleIntersectRay(data.triangles[i], losRay, distance, surfel, false)) return false;
    return true;
}

// ---- CudaTracer.cu:392-418 -----------------------------------------------------------------------
vec3 getAreaLightPoint(const ptss_area_light& light, const std::vector<ptss_triangle>& triangles, CurandState& state) {
    const float u1 = curand_uniform(&state);
    const float u2 = curand_uniform(&state);
    const float u3 = curand_uniform(&state);
    const float inverseTotal = 1 / (u1 + u2 + u3);
    const float weight0 = u1 * inverseTotal, weight1 = u2 * inverseTotal, weight2 = u3 * inverseTotal;
    const ptss_triangle& t = (curand_uniform(&state) > .5f) ? triangles[light.triangleIdx] : triangles[light.triangleIdx + 1];
    return (t.vertex0 * weight0 + t.vertex1 * weight1) + t.vertex2 * weight2;
}

// ---- CudaTracer.cu:345-390 -----------------------------------------------------------------------
inline void addLambertTerm(vec3& radiance, const vec3& normal, const vec3& w_i, const vec3& power, float distance2,
                           const ptss_material& material) {
    const vec3 L_i = power / (float)(4 * M_PI_F * distance2);
    const float cosI = ptm::max(0.0f, dot(normal, w_i));
    radiance.x += cosI * L_i.x * material.diffuseColor.x * material.diffAvg * INVERSE_PI;
    radiance.y += cosI * L_i.y * material.diffuseColor.y * material.diffAvg * INVERSE_PI;
    radiance.z += cosI * L_i.z * material.diffuseColor.z * material.diffAvg * INVERSE_PI;
}

vec3 shade(const RendererData& data, const SurfaceElement& surfel, const ptss_material& material, CurandState& state) {
    vec3 w_i;
    float distance2;
    vec3 radiance = v3(0, 0, 0);
    for (size_t i = 0; i < data.pointLights.size(); i++) {
        const ptss_point_light& light = data.pointLights[i];
        if (lineOfSight(data, surfel.normal, surfel.point, light.position, w_i, distance2))
            addLambertTerm(radiance, surfel.normal, w_i, light.power, distance2, material);
    }
    for (size_t i = 0; i < data.areaLights.size(); i++) {
        const ptss_area_light& light = data.areaLights[i];
        const vec3 point = getAreaLightPoint(light, data.triangles, state);
        if (lineOfSight(data, surfel.normal, surfel.point, point, w_i, distance2))
            addLambertTerm(radiance, surfel.normal, w_i, light.power, distance2, material);
    }
    return radiance;
}

// ---- CudaTracer.cu:208-318 -----------------------------------------------------------------------
vec3 computeIndirectRadianceAndScatter(Ray& ray, const SurfaceElement& surfel, const ptss_material& material,
                                       float cosI, CurandState& localState) {
    float r = curand_uniform(&localState);

    if (material.diffAvg > 0.0f) {
        r -= material.diffAvg;
        if (r < 0.0f) {
            ray.origin = surfel.point + RAY_BUMP_EPSILON * surfel.normal;
            ray.direction = randomDirectionLambert(surfel.normal, localState);
            return material.diffuseColor;
        }
    }

    float n1, n2, n, sinT2;
    computeSinT2AndRefractiveIndexes(material.indexOfRefraction, cosI, sinT2, n1, n2, n);
    const float fresnelReflective = computeFresnelForReflectance(cosI, sinT2, n1, n2, n);

    if (material.specAvg > 0.0f) {
        if (material.flags & PTSS_MAT_FLAG_PURE_REFLECTION)
            r -= material.specAvg;
        else
            r -= material.specAvg * fresnelReflective;

        if (r < 0.0f) {
            if (material.flags & PTSS_MAT_FLAG_COOK_TORRANCE) {
                const vec3 beckmannNormal = randomDirectionBeckmann(surfel.normal, material.roughness, localState);
                const vec3 incident = ray.direction;
                reflRay(ray, surfel.point, beckmannNormal);

                const vec3 half = normalize(ray.direction - incident);
                const float nh = ptm::abs(dot(surfel.normal, half));
                const float nl = ptm::abs(dot(surfel.normal, ray.direction));
                const float vh = ptm::abs(dot(incident, half));
                const float nv = ptm::abs(cosI);
                const float geometric = ptm::min(ptm::min(1.0f, 2 * nh * nl / vh), 2 * nh * nv / vh);
                return material.specularColor * geometric / nv;
            }
            reflRay(ray, surfel, cosI);
            if (material.specularExponent != ptm::inf())
                ray.direction = randomDirectionPhong(ray.direction, material.specularExponent, localState);
            return material.specularColor;
        }
    }

    if (material.refrAvg > 0.0f) {
        const float fresnelRefractive = 1.0f - fresnelReflective;
        r -= material.refrAvg * fresnelRefractive;
        if (r < 0.0f) {
            refrRay(ray, surfel, cosI, sinT2, n);
            return v3(1, 1, 1);
        }
    }

    ray.active = false;
    return v3(0, 0, 0);
}

// ---- CudaTracer.cu:321-343, generalised to W != H per SURVEY.md §9.5 ------------------------------
Ray computeEyeRay(int x, int y, int width, int height, const ptss_camera& camera, CurandState& state) {
    const float aspectRatio = (float)height / (float)width;
    const float inverseW = 1.0f / width;
    const float inverseH = 1.0f / height;
    const float jitteredX = x + curand_uniform(&state);
    const float jitteredY = y + curand_uniform(&state);
    const float s = -2 * ptm::tan(camera.fieldOfView * 0.5f);
    const vec3 start =
        v3(((jitteredX * inverseW) - 0.5f) * s, 1 * ((jitteredY * inverseH) - 0.5f) * s * aspectRatio, 1.0f) * camera.zNear;
    return makeRay(camera.position, normalize(rotate(camera.rotation, start)));
}

// ---- CudaTracer.cu:106-206, one thread ------------------------------------------------------------
void pathTraceOne(const RendererData& data, Ray& rayInOut, CurandState& stateInOut, bool isLastIteration) {
    CurandState localState = stateInOut;
    Ray ray = rayInOut;

    float distance = ptm::inf();
    SurfaceElement surfel;
    memset(&surfel, 0, sizeof(surfel));
    char intersection = 0;

    for (size_t i = 0; i < data.spheres.size(); i++)
        if (sphereIntersectRay(data.spheres[i], ray, distance, surfel)) intersection = 1;
    for (size_t i = 0; i < data.triangles.size(); i++)
        if (triangleIntersectRay(data.triangles[i], ray, distance, surfel)) intersection = 1;

    if (intersection) {
        const float cosI = dot(-ray.direction, surfel.normal);
        const ptss_material material = data.materials[surfel.materialIdx];

        vec3 directRadiance = v3(0, 0, 0);
        directRadiance = directRadiance + material.emmitance;

        const bool inside = cosI <= 0.0f;
        if (!inside) directRadiance = directRadiance + shade(data, surfel, material, localState);

        const vec3 indirectRadiance =
            isLastIteration ? v3(1, 1, 1) : computeIndirectRadianceAndScatter(ray, surfel, material, cosI, localState);

        if (inside) {
            ray.radiance1 = ray.radiance1 * v3(ptm::exp(-distance * material.absorption.x),
                                               ptm::exp(-distance * material.absorption.y),
                                               ptm::exp(-distance * material.absorption.z));
        }
        ray.radiance0 = ray.radiance0 + ray.radiance1 * directRadiance;
        ray.radiance1 = ray.radiance1 * indirectRadiance;
    } else {
        ray.radiance0 = ray.radiance0 + data.defaultColor * ray.radiance1;
        ray.active = false;
    }

    stateInOut = localState;
    rayInOut = ray;
}

// ---- CudaTracer.cu:72-85: one channel of one sample -> 8-bit ---------------------------------------
inline uint32_t quantizeSample(float radiance) {
    float v = ptm::clamp(radiance, 0.0f, 1.0f);
    v = ptm::pow(v, GAMMA_CORRECTION);
    v = ptm::clamp(255 * v + 0.5f, 0.f, 255.f);
    return (v == v) ? (uint32_t)v : 0u;  // NaN -> 0 (CUDA's float->uint of NaN)
}

}  // namespace

// =================================================================================================
struct oracle_ctx {
    RendererData data;
    int width, height;
    ptss_camera camera;
    unsigned maxIterations;
    bool usePathTracer;
    bool resetTicksThisFrame;
    int lastResetTick;
    int literalSlotRng;  // fidelity probe: slot-bound RNG + numRays/96 truncation
    int samples;         // extension: S independent sample lanes per pixel per frame; 1 = the reference
    std::vector<Ray> rays;
    std::vector<CurandState> curandStates;
    std::vector<uint32_t> totalPixelColors;  // uint3 per pixel
    std::vector<float> floatSum;             // linear sum of radiance0 per STREAM (lane * N + pixel) (extra, §9.1)
    std::vector<float> floatSumPixel;        // the same summed over lanes in lane order (filled on request)
    std::vector<float> lastRadiance0;        // radiance0 of the last frame, by pixel
    std::vector<uint32_t> liveCounts;        // rays entering each bounce of the last frame
    uint64_t totalRayBounces;
};

extern "C" {

oracle_ctx* oracle_create(const ptss_scene_desc* scene, int width, int height, unsigned long long seed,
                          unsigned maxIterations, int literalSlotRng, int samplesPerPass) {
    if (!scene || width <= 0 || height <= 0 || samplesPerPass < 1 || samplesPerPass > 64) return nullptr;
    if (literalSlotRng && samplesPerPass != 1) return nullptr;
    oracle_ctx* c = new oracle_ctx();
    c->data.defaultColor = scene->defaultColor;
    c->data.spheres.assign(scene->spheres, scene->spheres + scene->numSpheres);
    c->data.triangles.assign(scene->triangles, scene->triangles + scene->numTriangles);
    c->data.materials.assign(scene->materials, scene->materials + scene->numMaterials);
    c->data.pointLights.assign(scene->pointLights, scene->pointLights + scene->numPointLights);
    c->data.areaLights.assign(scene->areaLights, scene->areaLights + scene->numAreaLights);
    c->width = width;
    c->height = height;
    c->camera.rotation = q4(1, 0, 0, 0);  // Camera(), RenderStructs.h:51-52
    c->camera.position = v3(0, 0, 0);
    c->camera.zNear = -0.1f;
    c->camera.zFar = -100.0f;
    c->camera.fieldOfView = M_PI_F / 2.0f;
    c->maxIterations = maxIterations;  // reference default 15, CudaTracer.h:39
    c->usePathTracer = true;
    c->resetTicksThisFrame = true;  // CudaTracer.cu:717
    c->lastResetTick = 0;
    c->literalSlotRng = literalSlotRng;
    c->samples = samplesPerPass;
    const size_t n = (size_t)width * height;
    const size_t m = n * (size_t)samplesPerPass;  // streams = rays per frame
    c->rays.resize(m);
    c->curandStates.resize(m);
    c->totalPixelColors.assign(3 * n, 0u);
    c->floatSum.assign(3 * m, 0.0f);
    c->floatSumPixel.assign(3 * n, 0.0f);
    c->lastRadiance0.assign(3 * m, 0.0f);
    c->totalRayBounces = 0;
    (void)sequencePowers();
    // curandSetupKernel, CudaTracer.cu:22-29: same seed, sequence = slot
#pragma omp parallel for schedule(static)
    // stream (lane l, pixel p) is stored at l * N + p and owns subsequence p * S + l
    for (long i = 0; i < (long)m; ++i) {
        const long l = i / (long)n, pix = i % (long)n;
        curand_init(seed, (uint32_t)(pix * samplesPerPass + l), &c->curandStates[i]);
    }
    return c;
}

void oracle_destroy(oracle_ctx* c) { delete c; }

// OpenMP team size for every parallel loop of this library (bench.py: the box's CPU share)
void oracle_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}
int oracle_max_threads(void) { return omp_get_max_threads(); }

void oracle_set_camera(oracle_ctx* c, const ptss_camera* cam) {
    c->camera = *cam;
    c->resetTicksThisFrame = true;  // CudaTracer.cu:782-785
}
void oracle_set_mode(oracle_ctx* c, int usePathTracer) {
    c->usePathTracer = usePathTracer != 0;
    c->resetTicksThisFrame = true;  // CudaTracer.cu:763-764
}
void oracle_set_max_iterations(oracle_ctx* c, unsigned maxIterations) { c->maxIterations = maxIterations; }
void oracle_request_reset(oracle_ctx* c) { c->resetTicksThisFrame = true; }

// generateFrame, CudaTracer.cu:587-647. `pixels` is a HOST buffer of width*height uchar4 here.
void oracle_generate_frame(oracle_ctx* c, ptss_uchar4* pixels, int ticks) {
    const int W = c->width, H = c->height;
    const long N = (long)W * H;
    const int S = c->samples;
    const long M = N * S;

    if (c->resetTicksThisFrame) {  // :602-608 + clearPixels :31-49
        c->lastResetTick = ticks;
        memset(pixels, 0, sizeof(ptss_uchar4) * N);
        std::fill(c->totalPixelColors.begin(), c->totalPixelColors.end(), 0u);
        std::fill(c->floatSum.begin(), c->floatSum.end(), 0.0f);
        c->resetTicksThisFrame = false;
    }

    // computeEyeRaysKernel :51-61
#pragma omp parallel for schedule(static)
    for (long slot = 0; slot < M; ++slot) {
        const long offset = slot % N;
        const int x = (int)(offset % W), y = (int)(offset / W);
        Ray ray = computeEyeRay(x, y, W, H, c->camera, c->curandStates[slot]);
        ray.pixelOffset = (int)offset;
        ray.lane = (int)(slot / N);
        c->rays[slot] = ray;
    }

    long numRays = M;
    const unsigned numIterations = c->usePathTracer ? c->maxIterations : 1;  // :620
    c->liveCounts.assign(numIterations, 0u);
    std::vector<Ray> scratch;
    for (unsigned i = 0; i < numIterations && numRays > 128; i++) {  // :622
        const bool isLast = (i == numIterations - 1);
        const long launched = c->literalSlotRng ? (numRays / 96) * 96 : numRays;  // :623
        c->liveCounts[i] = (uint32_t)launched;
        c->totalRayBounces += (uint64_t)launched;
#pragma omp parallel for schedule(dynamic, 256)
        for (long slot = 0; slot < launched; ++slot) {
            Ray& ray = c->rays[slot];
            CurandState& st = c->literalSlotRng ? c->curandStates[slot] : c->curandStates[(long)ray.lane * N + ray.pixelOffset];
            pathTraceOne(c->data, ray, st, isLast);
        }
        if (!isLast) {  // :626-632 thrust::partition, made stable: actives first, both groups in slot order
            // (parallel: every thread counts the actives of its own contiguous chunk, a prefix over the threads gives each chunk
            // its two destinations, then every thread moves its chunk — the order inside and between chunks is kept)
            scratch.resize((size_t)numRays);
            const int T = omp_get_max_threads();
            std::vector<long> actives((size_t)T + 1, 0), headOf((size_t)T + 1, 0), tailOf((size_t)T + 1, 0);
            long head = 0;
#pragma omp parallel num_threads(T)
            {
                const int t = omp_get_thread_num(), nt = omp_get_num_threads();
                const long lo = numRays * t / nt, hi = numRays * (t + 1) / nt;
                long mine = 0;
                for (long k = lo; k < hi; ++k) {
                    scratch[k] = c->rays[k];
                    mine += scratch[k].active ? 1 : 0;
                }
                actives[t] = mine;
#pragma omp barrier
#pragma omp single
                {
                    long total = 0;
                    for (int q = 0; q < nt; ++q) total += actives[q];
                    long h = 0, tl = total;
                    for (int q = 0; q < nt; ++q) {
                        const long qlo = numRays * q / nt, qhi = numRays * (q + 1) / nt;
                        headOf[q] = h;
                        tailOf[q] = tl;
                        h += actives[q];
                        tl += (qhi - qlo) - actives[q];
                    }
                    head = total;
                }   // (implicit barrier)
                long h = headOf[t], tl = tailOf[t];
                for (long k = lo; k < hi; ++k) {
                    if (scratch[k].active) c->rays[h++] = scratch[k];
                    else c->rays[tl++] = scratch[k];
                }
            }
            numRays = head;
        }
    }

    // writeToPixelsKernel :63-104 over all slots: every sample is tone-mapped on its own, then summed
    const int sample = ticks - c->lastResetTick;
    const float inverseTicks = 1.f / (float)(S * (sample + 1));  // S = 1: 1.f / (ticks + 1), :94
    // (parallel over slots: a stream's float sum and last radiance belong to one slot; the integer sums of a pixel receive
    // S samples from S slots — integer adds commute, so atomic adds give the sequential loop's sums exactly)
#pragma omp parallel for schedule(static)
    for (long slot = 0; slot < M; ++slot) {
        const Ray& ray = c->rays[slot];
        const long p = ray.pixelOffset, stream = (long)ray.lane * N + p;
        const float rad[3] = {ray.radiance0.x, ray.radiance0.y, ray.radiance0.z};
        for (int ch = 0; ch < 3; ++ch) {
            const uint32_t q = quantizeSample(rad[ch]);
            if (S == 1) {
                c->totalPixelColors[3 * p + ch] += q;
            } else {
#pragma omp atomic
                c->totalPixelColors[3 * p + ch] += q;
            }
            c->floatSum[3 * stream + ch] += rad[ch];
            c->lastRadiance0[3 * stream + ch] = rad[ch];
        }
    }
#pragma omp parallel for schedule(static)
    for (long p = 0; p < N; ++p) {
        pixels[p].x = (unsigned char)(c->totalPixelColors[3 * p + 0] * inverseTicks + 0.5f);
        pixels[p].y = (unsigned char)(c->totalPixelColors[3 * p + 1] * inverseTicks + 0.5f);
        pixels[p].z = (unsigned char)(c->totalPixelColors[3 * p + 2] * inverseTicks + 0.5f);
        pixels[p].w = 255;
    }
}

const uint32_t* oracle_accumulator(const oracle_ctx* c) { return c->totalPixelColors.data(); }
const float* oracle_float_sum(oracle_ctx* c) {  // per pixel: the per-stream sums added in lane order 0..S-1
    const long N = (long)c->width * c->height;
    for (long k = 0; k < 3 * N; ++k) {
        float acc = 0.0f;
        for (int l = 0; l < c->samples; ++l) acc = acc + c->floatSum[3 * (long)l * N + k];
        c->floatSumPixel[k] = acc;
    }
    return c->floatSumPixel.data();
}
const float* oracle_last_radiance0(const oracle_ctx* c) { return c->lastRadiance0.data(); }
int oracle_live_counts(const oracle_ctx* c, uint32_t* out, int cap) {
    const int n = (int)c->liveCounts.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = c->liveCounts[i];
    return n;
}
unsigned long long oracle_total_ray_bounces(const oracle_ctx* c) { return c->totalRayBounces; }
void oracle_rng_state(const oracle_ctx* c, long pixel, int lane, uint32_t* out6) {
    const CurandState& s = c->curandStates[(long)lane * c->width * c->height + pixel];
    for (int i = 0; i < 5; ++i) out6[i] = s.v[i];
    out6[5] = s.d;
}

// ---- single-function probes for the known-answer tests (tests/test_oracle_kat.py) -----------------
// ray6 = origin xyz, direction xyz. Returns hit flag; out = distance, point xyz, normal xyz, materialIdx.
int oracle_probe_sphere(const ptss_sphere* sp, const float* ray6, float maxDistance, float* out8) {
    Ray r = makeRay(v3(ray6[0], ray6[1], ray6[2]), v3(ray6[3], ray6[4], ray6[5]));
    SurfaceElement se;
    memset(&se, 0, sizeof(se));
    float d = maxDistance;
    const int hit = sphereIntersectRay(*sp, r, d, se) ? 1 : 0;
    out8[0] = d; out8[1] = se.point.x; out8[2] = se.point.y; out8[3] = se.point.z;
    out8[4] = se.normal.x; out8[5] = se.normal.y; out8[6] = se.normal.z; out8[7] = (float)se.materialIdx;
    return hit;
}
int oracle_probe_triangle(const ptss_triangle* tri, const float* ray6, float maxDistance, float* out8) {
    Ray r = makeRay(v3(ray6[0], ray6[1], ray6[2]), v3(ray6[3], ray6[4], ray6[5]));
    SurfaceElement se;
    memset(&se, 0, sizeof(se));
    float d = maxDistance;
    const int hit = triangleIntersectRay(*tri, r, d, se) ? 1 : 0;
    out8[0] = d; out8[1] = se.point.x; out8[2] = se.point.y; out8[3] = se.point.z;
    out8[4] = se.normal.x; out8[5] = se.normal.y; out8[6] = se.normal.z; out8[7] = (float)se.materialIdx;
    return hit;
}
float oracle_probe_fresnel(float refrIndex, float cosI) {
    float sinT2, n1, n2, n;
    computeSinT2AndRefractiveIndexes(refrIndex, cosI, sinT2, n1, n2, n);
    return computeFresnelForReflectance(cosI, sinT2, n1, n2, n);
}
void oracle_probe_rotate_y_to(const float* target3, const float* v3in, float* out3) {
    const vec3 r = rotate(rotateVectorToVector(v3(0, 1, 0), v3(target3[0], target3[1], target3[2])),
                          v3(v3in[0], v3in[1], v3in[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// kind: 0 Lambert(normal), 1 Phong(w_o, param=exponent), 2 Beckmann(normal, param=roughness)
void oracle_probe_sampler(int kind, const float* axis3, float param, unsigned long long seed, int n, float* out3n) {
    CurandState st;
    curand_init(seed, 0, &st);
    const vec3 a = v3(axis3[0], axis3[1], axis3[2]);
    for (int i = 0; i < n; ++i) {
        vec3 d;
        if (kind == 0) d = randomDirectionLambert(a, st);
        else if (kind == 1) d = randomDirectionPhong(a, param, st);
        else d = randomDirectionBeckmann(a, param, st);
        out3n[3 * i] = d.x; out3n[3 * i + 1] = d.y; out3n[3 * i + 2] = d.z;
    }
}
// shade() of one surfel against a scene; out3 = radiance. RNG: curand_init(seed, 0).
void oracle_probe_shade(const oracle_ctx* c, const float* point3, const float* normal3, int materialIdx,
                        unsigned long long seed, float* out3) {
    CurandState st;
    curand_init(seed, 0, &st);
    SurfaceElement se;
    se.point = v3(point3[0], point3[1], point3[2]);
    se.normal = v3(normal3[0], normal3[1], normal3[2]);
    se.materialIdx = materialIdx;
    const vec3 r = shade(c->data, se, c->data.materials[materialIdx], st);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
unsigned oracle_probe_quantize(float radiance) { return quantizeSample(radiance); }
void oracle_probe_rng(unsigned long long seed, unsigned sequence, int n, uint32_t* state6, uint32_t* raw, float* uni) {
    CurandState st;
    curand_init(seed, sequence, &st);
    if (state6) {
        for (int i = 0; i < 5; ++i) state6[i] = st.v[i];
        state6[5] = st.d;
    }
    for (int i = 0; i < n; ++i) {
        CurandState b = st;
        const uint32_t r = curand(st);
        if (raw) raw[i] = r;
        if (uni) uni[i] = curand_uniform(&b);
    }
}
// one eye ray: out6 = origin, direction; RNG state of `pixel` under `seed`.
void oracle_probe_eye_ray(int x, int y, int width, int height, const ptss_camera* cam, unsigned long long seed,
                          float* out6) {
    CurandState st;
    curand_init(seed, (uint32_t)(y * width + x), &st);
    const Ray r = computeEyeRay(x, y, width, height, *cam, st);
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
}

}  // extern "C"
