// ptss_kernels.hip — the hot path as hand-written HIP for gfx950 (CDNA4, 64-lane waves).
//
// Kernels (reference kernels they replace, paths relative to /root/reference/CudaTracer/):
//   rngInitKernel   <- curandSetupKernel            CudaTracer.cu:22-29
//   clearKernel     <- clearPixels                  CudaTracer.cu:31-49
//   eyeRaysKernel   <- computeEyeRaysKernel         CudaTracer.cu:51-61, 321-343
//   bounceKernel    <- pathTraceKernel + thrust::partition + the per-ray part of writeToPixelsKernel
//                                                   CudaTracer.cu:106-206, :629, :63-104
//   flushKernel     <- writeToPixelsKernel for rays still alive when the loop guard stops the frame
//                                                   CudaTracer.cu:622, :63-104
//
// Design: one ray per lane; ray state in SoA planes (coalesced 256-B wave accesses);
// the whole scene staged once per workgroup into LDS and read by broadcast; live rays are
// compacted in the same kernel that traces them — 64-bit __ballot + popcount lane rank + one
// atomic per wave on a device-resident counter, so the host never reads a ray count inside a
// frame; a ray that ends (miss, absorbed, last bounce) tone-maps and adds its sample into the
// integer accumulator right there and parks its XORWOW state back in the per-pixel home slot.
// No MFMA: there is no dense contraction in this path.
//
// Arithmetic mirrors oracle/oracle.cpp operation for operation (ptmath.h; -ffp-contract=off).
#include "ptss_device.h"

using namespace ptv;

namespace ptss {
namespace {

__device__ __forceinline__ float asF(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t asU(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ vec3 xyz(float4 v) { return vec3{v.x, v.y, v.z}; }

struct PixelCoord {
    int x, gy;
    uint32_t globalIndex;
};

__device__ __forceinline__ PixelCoord locate(const TileMap& t, uint32_t local) {
    const int lx = (int)(local % (uint32_t)t.width);
    const int ly = (int)(local / (uint32_t)t.width);
    const int band = ly / t.bandRows, within = ly % t.bandRows;
    PixelCoord p;
    p.x = lx;
    p.gy = (band * t.world + t.rank) * t.bandRows + within;
    p.globalIndex = (uint32_t)p.gy * (uint32_t)t.width + (uint32_t)lx;
    return p;
}

struct RayRegs {
    vec3 o, d, L0, T;
    uint32_t pix;
    ptrng::State rng;
    bool active;
};

__device__ __forceinline__ void loadRay(const float* __restrict__ pool, uint32_t cap, uint32_t i, RayRegs& r) {
    r.o = vec3{pool[kOx * cap + i], pool[kOy * cap + i], pool[kOz * cap + i]};
    r.d = vec3{pool[kDx * cap + i], pool[kDy * cap + i], pool[kDz * cap + i]};
    r.L0 = vec3{pool[kL0x * cap + i], pool[kL0y * cap + i], pool[kL0z * cap + i]};
    r.T = vec3{pool[kTx * cap + i], pool[kTy * cap + i], pool[kTz * cap + i]};
    r.pix = asU(pool[kPix * cap + i]);
    r.rng.v[0] = asU(pool[kR0 * cap + i]);
    r.rng.v[1] = asU(pool[kR1 * cap + i]);
    r.rng.v[2] = asU(pool[kR2 * cap + i]);
    r.rng.v[3] = asU(pool[kR3 * cap + i]);
    r.rng.v[4] = asU(pool[kR4 * cap + i]);
    r.rng.d = asU(pool[kRd * cap + i]);
    r.active = true;
}

__device__ __forceinline__ void storeRay(float* __restrict__ pool, uint32_t cap, uint32_t i, const RayRegs& r) {
    pool[kOx * cap + i] = r.o.x;  pool[kOy * cap + i] = r.o.y;  pool[kOz * cap + i] = r.o.z;
    pool[kDx * cap + i] = r.d.x;  pool[kDy * cap + i] = r.d.y;  pool[kDz * cap + i] = r.d.z;
    pool[kL0x * cap + i] = r.L0.x; pool[kL0y * cap + i] = r.L0.y; pool[kL0z * cap + i] = r.L0.z;
    pool[kTx * cap + i] = r.T.x;  pool[kTy * cap + i] = r.T.y;  pool[kTz * cap + i] = r.T.z;
    pool[kPix * cap + i] = asF(r.pix);
    pool[kR0 * cap + i] = asF(r.rng.v[0]);
    pool[kR1 * cap + i] = asF(r.rng.v[1]);
    pool[kR2 * cap + i] = asF(r.rng.v[2]);
    pool[kR3 * cap + i] = asF(r.rng.v[3]);
    pool[kR4 * cap + i] = asF(r.rng.v[4]);
    pool[kRd * cap + i] = asF(r.rng.d);
}

// ---- Sphere::intersectRay, Primitives.h:107-175. sp = {centre, radius^2}. ---------------------
// Returns the accepted distance in t; `limit` is the running `distance`.
__device__ __forceinline__ bool sphereTest(float4 sp, vec3 o, vec3 d, float limit, float& t) {
    const vec3 v = o - xyz(sp);
    const float b = dot(d, v) * 2;
    const float c = dot(v, v) - sp.w;
    float disc = (b * b) - 4 * c;
    if (disc < 0) return false;
    disc = ptm::sqrt(disc);
    float t0 = (-b + disc) * 0.5f;
    float t1 = (-b - disc) * 0.5f;
    if (t0 < 0 && t1 < 0) return false;
    if (t0 > t1) {
        const float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    const float cand = (t0 < 0) ? t1 : t0;
    if (cand > limit) return false;
    t = cand;
    return true;
}

// ---- Triangle::intersectRay, Primitives.h:25-83. v0/e1/e2 from the staged scene. ---------------
__device__ __forceinline__ bool triangleTest(vec3 v0, vec3 e1, vec3 e2, vec3 o, vec3 d, float limit, float& t,
                                             float& w0, float& w1, float& w2) {
    const vec3 q = cross(d, e2);
    const float det = dot(e1, q);
    if (ptm::abs(det) <= 1e-7f) return false;
    const float inverseDet = 1 / det;
    const vec3 s = o - v0;
    const vec3 r = cross(s, e1);
    const float dist = dot(e2, r) * inverseDet;
    if ((dist <= 0.0f) || (dist > limit)) return false;
    const float b1 = dot(s, q) * inverseDet;
    const float b2 = dot(d, r) * inverseDet;
    const float b0 = 1.0f - (b1 + b2);
    if ((b0 < 0) || (b1 < 0) || (b2 < 0)) return false;
    t = dist;
    w0 = b0;
    w1 = b1;
    w2 = b2;
    return true;
}

// ---- lineOfSight, CudaTracer.cu:420-455 (any-hit; order-independent because it returns at the
// first accepted primitive and no test depends on another). Per-lane early return: the wave leaves
// the loops as soon as every lane that entered is occluded. ---------------------------------------
__device__ __forceinline__ bool lineOfSight(const float4* sc, const SceneLayout& L, vec3 normal, vec3 point0,
                                            vec3 point1, vec3& w_i, float& distance2) {
    const vec3 offset = point1 - point0;
    distance2 = dot(offset, offset);
    float distance = ptm::sqrt(distance2);
    w_i = offset / distance;
    const vec3 lo = point0 + (ptm::kRayBump * normal);
    distance -= 2 * ptm::kRayBump;
    float t, w0, w1, w2;
    for (int i = 0; i < L.numSpheres; ++i)
        if (sphereTest(sc[L.offSphere + i], lo, w_i, distance, t)) return false;
    for (int i = 0; i < L.numTriangles; ++i) {
        const float4* tr = sc + L.offTri + 3 * i;
        if (triangleTest(xyz(tr[0]), xyz(tr[1]), xyz(tr[2]), lo, w_i, distance, t, w0, w1, w2)) return false;
    }
    return true;
}

// one light's Lambert term, CudaTracer.cu:360-366 / :379-385
__device__ __forceinline__ void addLambertTerm(vec3& radiance, vec3 normal, vec3 w_i, vec3 power, float distance2,
                                               float4 diffuse /* colour, diffAvg */) {
    const vec3 L_i = power / (float)(4 * ptm::kPi * distance2);
    const float cosI = ptm::max(0.0f, dot(normal, w_i));
    radiance.x += cosI * L_i.x * diffuse.x * diffuse.w * ptm::kInvPi;
    radiance.y += cosI * L_i.y * diffuse.y * diffuse.w * ptm::kInvPi;
    radiance.z += cosI * L_i.z * diffuse.z * diffuse.w * ptm::kInvPi;
}

// ---- shade, CudaTracer.cu:345-390 + getAreaLightPoint :392-418 --------------------------------
__device__ __forceinline__ vec3 shade(const float4* sc, const SceneLayout& L, vec3 point, vec3 normal, float4 diffuse,
                                      ptrng::State& rng) {
    vec3 radiance = v3(0, 0, 0);
    vec3 w_i;
    float distance2;
    for (int i = 0; i < L.numPointLights; ++i) {
        const float4 pos = sc[L.offPointLight + 2 * i], pw = sc[L.offPointLight + 2 * i + 1];
        if (lineOfSight(sc, L, normal, point, xyz(pos), w_i, distance2))
            addLambertTerm(radiance, normal, w_i, xyz(pw), distance2, diffuse);
    }
    for (int i = 0; i < L.numAreaLights; ++i) {
        const float4 light = sc[L.offAreaLight + i];
        const float u1 = ptrng::uniform(rng);
        const float u2 = ptrng::uniform(rng);
        const float u3 = ptrng::uniform(rng);
        const float inverseTotal = 1 / (u1 + u2 + u3);
        const float weight0 = u1 * inverseTotal, weight1 = u2 * inverseTotal, weight2 = u3 * inverseTotal;
        const int tri = (int)asU(light.w) + ((ptrng::uniform(rng) > .5f) ? 0 : 1);
        const vec3 a = xyz(sc[L.offTri + 3 * tri]);
        const vec3 b = xyz(sc[L.offTriVert + 2 * tri]);
        const vec3 c = xyz(sc[L.offTriVert + 2 * tri + 1]);
        const vec3 lightPoint = (a * weight0 + b * weight1) + c * weight2;
        if (lineOfSight(sc, L, normal, point, lightPoint, w_i, distance2))
            addLambertTerm(radiance, normal, w_i, xyz(light), distance2, diffuse);
    }
    return radiance;
}

// CudaTracer.cu:579-585
__device__ __forceinline__ quat rotateVectorToVector(vec3 source, vec3 target) {
    const vec3 axis = cross(source, target);
    return normalize(q4(1.0f + dot(source, target), axis.x, axis.y, axis.z));
}

// shared tail of the Lambert / Phong samplers, CudaTracer.cu:536-544, 550-558
__device__ __forceinline__ vec3 lobeSample(vec3 axis, float theta, float y) {
    const float r = ptm::sqrt(1 - y * y);
    float sn, cs;
    ptm::sincos(theta, sn, cs);
    return rotate(rotateVectorToVector(v3(0, 1, 0), axis), v3(r * cs, y, r * sn));
}

// ---- computeIndirectRadianceAndScatter, CudaTracer.cu:208-318 ---------------------------------
__device__ __forceinline__ vec3 scatter(const float4* mat, RayRegs& ray, vec3 point, vec3 normal, float cosI) {
    const float4 mDiffuse = mat[0];   // diffuseColor, diffAvg
    const float4 mSpecular = mat[1];  // specularColor, specAvg
    const float4 mMisc = mat[4];      // specularExponent, indexOfRefraction, flags
    const float refrAvg = mat[2].w;
    const int flags = (int)asU(mMisc.z);

    float r = ptrng::uniform(ray.rng);

    if (mDiffuse.w > 0.0f) {
        r -= mDiffuse.w;
        if (r < 0.0f) {
            ray.o = point + ptm::kRayBump * normal;
            const float theta = ptrng::uniform(ray.rng) * 2 * ptm::kPi;
            const float s = ptrng::uniform(ray.rng);
            ray.d = lobeSample(normal, theta, ptm::sqrt(s));  // randomDirectionLambert :533-545
            return xyz(mDiffuse);
        }
    }

    // computeSinT2AndRefractiveIndexes :474-494 (flips cosI when inside)
    float n1, n2;
    if (cosI > 0) {
        n2 = mMisc.y;
        n1 = 1.0f;
    } else {
        cosI = -cosI;
        n1 = mMisc.y;
        n2 = 1.0f;
    }
    const float n = n1 / n2;
    const float sinT2 = n * n * (1.0f - cosI * cosI);

    // computeFresnelForReflectance :457-472
    float fresnelReflective = 1.0f;
    if (!(sinT2 > 1.0f)) {
        const float cosT = ptm::sqrt(1.0f - sinT2);
        const float r_s = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
        const float r_p = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
        fresnelReflective = (r_s * r_s + r_p * r_p) * 0.5f;
    }

    if (mSpecular.w > 0.0f) {
        if (flags & PTSS_MAT_FLAG_PURE_REFLECTION)
            r -= mSpecular.w;
        else
            r -= mSpecular.w * fresnelReflective;

        if (r < 0.0f) {
            if (flags & PTSS_MAT_FLAG_COOK_TORRANCE) {
                // randomDirectionBeckmann :561-577
                const float roughness = mat[3].w;
                const float theta = ptm::atan(-roughness * roughness * ptm::log(1.0f - ptrng::uniform(ray.rng)));
                const float phi = ptrng::uniform(ray.rng) * 2 * ptm::kPi;
                float sinPhi, cosPhi, sinTheta, cosTheta;
                ptm::sincos(phi, sinPhi, cosPhi);
                ptm::sincos(theta, sinTheta, cosTheta);
                const vec3 m = v3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi);
                const vec3 beckmannNormal = rotate(rotateVectorToVector(v3(0, 1, 0), normal), m);

                const vec3 incident = ray.d;
                // reflRay(ray, point, normal) :505-514
                const float cosB = ptm::abs(dot(ray.d, beckmannNormal));
                ray.d = ray.d - (2 * (-cosB)) * beckmannNormal;
                ray.o = point + (beckmannNormal * ptm::kRayBump);

                const vec3 half = normalize(ray.d - incident);
                const float nh = ptm::abs(dot(normal, half));
                const float nl = ptm::abs(dot(normal, ray.d));
                const float vh = ptm::abs(dot(incident, half));
                const float nv = ptm::abs(cosI);
                const float geometric = ptm::min(ptm::min(1.0f, 2 * nh * nl / vh), 2 * nh * nv / vh);
                return xyz(mSpecular) * geometric / nv;
            }
            // reflRay(ray, surfel, cosI) :496-503
            ray.d = ray.d - (2 * (-cosI)) * normal;
            ray.o = point + (normal * ptm::kRayBump);
            if (mMisc.x != ptm::inf()) {  // randomDirectionPhong :547-559
                const float theta = ptrng::uniform(ray.rng) * 2 * ptm::kPi;
                const float s = ptrng::uniform(ray.rng);
                ray.d = lobeSample(ray.d, theta, ptm::pow(s, 1 / (mMisc.x + 1)));
            }
            return xyz(mSpecular);
        }
    }

    if (refrAvg > 0.0f) {
        const float fresnelRefractive = 1.0f - fresnelReflective;
        r -= refrAvg * fresnelRefractive;
        if (r < 0.0f) {
            // refrRay :516-531
            if (sinT2 > 1.0f) ray.active = false;
            const float cosT = ptm::sqrt(1.0f - sinT2);
            const vec3 w_o = normalize(n * ray.d + (n * cosI - cosT) * normal);
            ray.o = point + (w_o * ptm::kRayBump);
            ray.d = w_o;
            return v3(1, 1, 1);
        }
    }

    ray.active = false;
    return v3(0, 0, 0);
}

// one channel of writeToPixelsKernel, CudaTracer.cu:72-85
__device__ __forceinline__ uint32_t quantizeSample(float radiance) {
    float v = ptm::clamp(radiance, 0.0f, 1.0f);
    v = ptm::pow(v, ptm::kGamma);
    v = ptm::clamp(255 * v + 0.5f, 0.f, 255.f);
    return (v == v) ? (uint32_t)v : 0u;
}

// A path ended: writeToPixelsKernel for this ray (CudaTracer.cu:63-104) + park the RNG stream.
__device__ __forceinline__ void finishPath(const FrameBuffers& fb, const RayRegs& r) {
    const uint32_t p = r.pix;
    uint32_t* acc = fb.accum + 3u * p;
    const uint32_t tx = acc[0] + quantizeSample(r.L0.x);
    const uint32_t ty = acc[1] + quantizeSample(r.L0.y);
    const uint32_t tz = acc[2] + quantizeSample(r.L0.z);
    acc[0] = tx;
    acc[1] = ty;
    acc[2] = tz;
    if (fb.pixels) {
        ptss_uchar4 px;
        px.x = (unsigned char)(tx * fb.inverseTicks + 0.5f);
        px.y = (unsigned char)(ty * fb.inverseTicks + 0.5f);
        px.z = (unsigned char)(tz * fb.inverseTicks + 0.5f);
        px.w = 255;
        fb.pixels[p] = px;
    }
    if (fb.fsum) {
        float* fs = fb.fsum + 3u * p;
        fs[0] += r.L0.x;
        fs[1] += r.L0.y;
        fs[2] += r.L0.z;
    }
    const uint32_t cap = fb.capacity;
    fb.rngHome[0 * cap + p] = r.rng.v[0];
    fb.rngHome[1 * cap + p] = r.rng.v[1];
    fb.rngHome[2 * cap + p] = r.rng.v[2];
    fb.rngHome[3 * cap + p] = r.rng.v[3];
    fb.rngHome[4 * cap + p] = r.rng.v[4];
    fb.rngHome[5 * cap + p] = r.rng.d;
}

// ---- one thread of pathTraceKernel, CudaTracer.cu:106-206 -------------------------------------
template <bool kLast>
__device__ __forceinline__ void traceOne(const float4* sc, const SceneLayout& L, const FrameBuffers& fb, RayRegs& ray) {
    float distance = ptm::inf();
    int hitKind = 0, hitIdx = 0;  // 1 sphere, 2 triangle
    float w0 = 0, w1 = 0, w2 = 0;

    for (int i = 0; i < L.numSpheres; ++i) {
        float t;
        if (sphereTest(sc[L.offSphere + i], ray.o, ray.d, distance, t)) {
            distance = t;
            hitKind = 1;
            hitIdx = i;
        }
    }
    for (int i = 0; i < L.numTriangles; ++i) {
        const float4* tr = sc + L.offTri + 3 * i;
        float t, a0, a1, a2;
        if (triangleTest(xyz(tr[0]), xyz(tr[1]), xyz(tr[2]), ray.o, ray.d, distance, t, a0, a1, a2)) {
            distance = t;
            hitKind = 2;
            hitIdx = i;
            w0 = a0;
            w1 = a1;
            w2 = a2;
        }
    }

    if (hitKind == 0) {  // :193-198
        const vec3 dc = v3(fb.defaultColor[0], fb.defaultColor[1], fb.defaultColor[2]);
        ray.L0 = ray.L0 + dc * ray.T;
        ray.active = false;
        return;
    }

    // surfel of the winning primitive (Primitives.h:69-77, :98-105)
    const vec3 point = ray.o + ray.d * distance;
    vec3 normal;
    int materialIdx;
    if (hitKind == 1) {
        const float4 sp = sc[L.offSphere + hitIdx];
        normal = normalize(point - xyz(sp));
        materialIdx = reinterpret_cast<const int*>(sc + L.offSphereMat)[hitIdx];
    } else {
        const float4* nn = sc + L.offTriNormal + 3 * hitIdx;
        normal = (xyz(nn[0]) * w0 + xyz(nn[1]) * w1) + xyz(nn[2]) * w2;
        materialIdx = (int)asU(sc[L.offTri + 3 * hitIdx].w);
    }

    const float cosI = dot(-ray.d, normal);
    const float4* mat = sc + L.offMaterial + 5 * materialIdx;

    vec3 directRadiance = v3(0, 0, 0) + xyz(mat[3]);  // emmitance, :163
    const bool inside = cosI <= 0.0f;
    if (!inside) directRadiance = directRadiance + shade(sc, L, point, normal, mat[0], ray.rng);

    vec3 indirectRadiance = v3(1, 1, 1);
    if (!kLast) indirectRadiance = scatter(mat, ray, point, normal, cosI);

    if (inside) {  // Beer-Lambert, :179-185
        const float4 ab = mat[2];
        ray.T = ray.T * v3(ptm::exp(-distance * ab.x), ptm::exp(-distance * ab.y), ptm::exp(-distance * ab.z));
    }
    ray.L0 = ray.L0 + ray.T * directRadiance;
    ray.T = ray.T * indirectRadiance;
}

}  // namespace

// =================================================================================================
__global__ void rngInitKernel(uint32_t* __restrict__ rngHome, uint32_t capacity, TileMap tile, uint64_t seed,
                              const uint32_t* __restrict__ jumpTable) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = (uint32_t)tile.width * (uint32_t)tile.localRows;
    if (i >= n) return;
    const PixelCoord pc = locate(tile, i);
    ptrng::State s = ptrng::seeded(seed);
    ptrng::skip_subsequences(s, pc.globalIndex, jumpTable);
    for (int k = 0; k < 5; ++k) rngHome[k * capacity + i] = s.v[k];
    rngHome[5 * capacity + i] = s.d;
}

__global__ void clearKernel(FrameBuffers fb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fb.numPixels) return;
    fb.accum[3 * i] = 0;
    fb.accum[3 * i + 1] = 0;
    fb.accum[3 * i + 2] = 0;
    if (fb.fsum) {
        fb.fsum[3 * i] = 0;
        fb.fsum[3 * i + 1] = 0;
        fb.fsum[3 * i + 2] = 0;
    }
    if (fb.pixels) fb.pixels[i] = ptss_uchar4{0, 0, 0, 0};
}

__global__ void eyeRaysKernel(FrameBuffers fb, TileMap tile, EyeParams eye, int numBounces) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        fb.counts[0] = fb.numPixels;
        for (int b = 1; b <= numBounces; ++b) fb.counts[b] = 0;
    }
    if (i >= fb.numPixels) return;
    const PixelCoord pc = locate(tile, i);
    const uint32_t cap = fb.capacity;
    RayRegs r;
    for (int k = 0; k < 5; ++k) r.rng.v[k] = fb.rngHome[k * cap + i];
    r.rng.d = fb.rngHome[5 * cap + i];

    const float jitteredX = pc.x + ptrng::uniform(r.rng);
    const float jitteredY = pc.gy + ptrng::uniform(r.rng);
    const vec3 start = v3(((jitteredX * eye.invW) - 0.5f) * eye.s,
                          1 * ((jitteredY * eye.invH) - 0.5f) * eye.s * eye.aspect, 1.0f) *
                       eye.camera.zNear;
    r.o = eye.camera.position;
    r.d = normalize(rotate(eye.camera.rotation, start));
    r.L0 = v3(0, 0, 0);
    r.T = v3(1, 1, 1);
    r.pix = i;
    r.active = true;
    storeRay(fb.pool[0], cap, i, r);
}

// kSceneInLds = true : the scene blob is staged into LDS once per workgroup and read by broadcast
//                      (ds_read_b128, same address in every lane).
// kSceneInLds = false: the blob is read in place through wave-uniform addresses, which hipcc turns
//                      into scalar loads (s_load_dwordx4 -> SGPR operands, scalar cache); no staging,
//                      no LDS. Chosen per scene size at context creation (ptss_api.hip).
template <bool kLast, bool kSceneInLds>
__global__ __launch_bounds__(kBlock) void bounceKernel(FrameBuffers fb, const float4* __restrict__ sceneBlob,
                                                       SceneLayout L, int bounce) {
    extern __shared__ float4 lds[];
    const uint32_t n = fb.counts[bounce];
    if (n <= kMinLiveRays) return;  // loop guard, CudaTracer.cu:622 (device-side; same for every workgroup)

    // block-level compaction scratch lives behind the scene image (one LDS object, 16-B aligned)
    uint32_t* scratch = reinterpret_cast<uint32_t*>(lds + (kSceneInLds ? L.totalVec4 : 0));
    const float4* sc;
    if constexpr (kSceneInLds) {
        for (int k = threadIdx.x; k < L.totalVec4; k += kBlock) lds[k] = sceneBlob[k];
        __syncthreads();
        sc = lds;
    } else {
        sc = sceneBlob;
    }

    const float* __restrict__ in = fb.pool[bounce & 1];
    float* __restrict__ out = fb.pool[(bounce + 1) & 1];
    const uint32_t cap = fb.capacity;
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;

    // one tile per workgroup when the host's grid hint is right; grid-stride keeps any n correct
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        RayRegs ray;
        if (i < n) {
            loadRay(in, cap, i, ray);
            traceOne<kLast>(sc, L, fb, ray);
            alive = ray.active && !kLast;
            if (!alive) finishPath(fb, ray);
        }
        if constexpr (!kLast) {
            // stream compaction of the survivors: 64-bit ballot + lane rank inside the wave, the four
            // wave totals combined through LDS, ONE atomic per workgroup on the device-resident counter
            const unsigned long long live = __ballot(alive);
            const uint32_t rank = __popcll(live & ((1ull << lane) - 1ull));
            if (lane == 0) scratch[wave] = (uint32_t)__popcll(live);
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
                scratch[4] = total ? atomicAdd(&fb.counts[bounce + 1], total) : 0u;
            }
            __syncthreads();
            uint32_t slot = scratch[4] + rank;
            for (uint32_t w = 0; w < wave; ++w) slot += scratch[w];
            if (alive) storeRay(out, cap, slot, ray);
            __syncthreads();  // scratch is rewritten by the next tile
        }
    }
}

// After the last launched bounce: tone-map whatever the loop guard left alive (<= 128 rays), and
// add this frame's ray-bounce total to the running counter.
__global__ void flushKernel(FrameBuffers fb, int numBounces) {
    int stop = numBounces;
    for (int b = 0; b < numBounces; ++b)
        if (fb.counts[b] <= kMinLiveRays) {
            stop = b;
            break;
        }
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (int b = 0; b < stop; ++b) sum += fb.counts[b];
        *fb.totalRayBounces += sum;
    }
    const uint32_t n = fb.counts[stop];  // counts[numBounces] == 0 when the last bounce ran
    const uint32_t i = threadIdx.x;
    if (i < n) {
        RayRegs ray;
        loadRay(fb.pool[stop & 1], fb.capacity, i, ray);
        finishPath(fb, ray);
    }
}

// =================================================================================================
static inline unsigned blocksFor(uint32_t n, unsigned block) { return (n + block - 1) / block; }

hipError_t launchRngInit(hipStream_t st, uint32_t* rngHome, uint32_t capacity, TileMap tile, uint64_t seed,
                         const uint32_t* jumpTable) {
    const uint32_t n = (uint32_t)tile.width * (uint32_t)tile.localRows;
    hipLaunchKernelGGL(rngInitKernel, dim3(blocksFor(n, 256)), dim3(256), 0, st, rngHome, capacity, tile, seed, jumpTable);
    return hipGetLastError();
}

hipError_t launchClear(hipStream_t st, const FrameBuffers& fb) {
    hipLaunchKernelGGL(clearKernel, dim3(blocksFor(fb.numPixels, 256)), dim3(256), 0, st, fb);
    return hipGetLastError();
}

hipError_t launchEyeRays(hipStream_t st, const FrameBuffers& fb, TileMap tile, EyeParams eye, int numBounces) {
    hipLaunchKernelGGL(eyeRaysKernel, dim3(blocksFor(fb.numPixels, 256)), dim3(256), 0, st, fb, tile, eye, numBounces);
    return hipGetLastError();
}

template <bool kLast, bool kLds>
static hipError_t launchBounceT(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, const SceneLayout& layout,
                                int bounce, int gridBlocks) {
    const size_t lds = bounceLdsBytes(layout, kLds);
    hipLaunchKernelGGL((bounceKernel<kLast, kLds>), dim3(gridBlocks), dim3(kBlock), lds, st, fb, sceneBlob, layout, bounce);
    return hipGetLastError();
}

size_t bounceLdsBytes(const SceneLayout& layout, bool sceneInLds) {
    return ((sceneInLds ? (size_t)layout.totalVec4 : 0) + 2) * sizeof(float4);
}

hipError_t launchBounce(hipStream_t st, const FrameBuffers& fb, const float4* sceneBlob, SceneLayout layout, int bounce,
                        bool isLast, bool sceneInLds, int gridBlocks) {
    if (sceneInLds)
        return isLast ? launchBounceT<true, true>(st, fb, sceneBlob, layout, bounce, gridBlocks)
                      : launchBounceT<false, true>(st, fb, sceneBlob, layout, bounce, gridBlocks);
    return isLast ? launchBounceT<true, false>(st, fb, sceneBlob, layout, bounce, gridBlocks)
                  : launchBounceT<false, false>(st, fb, sceneBlob, layout, bounce, gridBlocks);
}

hipError_t launchFlush(hipStream_t st, const FrameBuffers& fb, int numBounces) {
    hipLaunchKernelGGL(flushKernel, dim3(1), dim3(kMinLiveRays), 0, st, fb, numBounces);
    return hipGetLastError();
}

int bounceOccupancyBlocksPerCU(const SceneLayout& layout, bool sceneInLds) {
    const size_t lds = bounceLdsBytes(layout, sceneInLds);
    int a = 0;
    hipError_t e = sceneInLds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, true>, kBlock, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bounceKernel<false, false>, kBlock, lds);
    return e == hipSuccess ? a : 0;
}

}  // namespace ptss
