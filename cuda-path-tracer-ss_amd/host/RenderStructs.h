// RenderStructs.h — host-side C++ faces of the boundary records (same names, same field order
// and defaults as the reference's CudaTracer/RenderStructs.h:24-121), layered on the plain-C
// layouts of include/ptss_types.h so that std::vector<T>::data() is directly a ptss_* array.
// glm is not available here; vec3/quat are the PODs plus the ptv:: operators of ptmath.h.
#pragma once
#include <cmath>
#include <cstddef>
#include "ptmath.h"
#include "ptss_types.h"

using ptv::quat;
using ptv::vec3;
using ptv::v3;

#ifndef PTSS_INFINITY
#define PTSS_INFINITY (__builtin_huge_valf())
#endif

#define MAT_FLAG_PURE_REFLECTION PTSS_MAT_FLAG_PURE_REFLECTION
#define MAT_FLAG_COOK_TORRANCE PTSS_MAT_FLAG_COOK_TORRANCE

// Ray as the reference declares it (RenderStructs.h:24-39). The device pools are SoA;
// this AoS record only exists for host-side inspection of one ray.
struct Ray {
    vec3 origin;
    vec3 direction;
    vec3 radiance0;
    vec3 radiance1;
    int pixelOffset;
    bool active;
    Ray(vec3 o, vec3 d) : origin(o), direction(d), radiance0(v3(0)), radiance1(v3(1)), pixelOffset(0), active(true) {}
    explicit Ray(bool alive = true) : origin(v3(0)), direction(v3(0)), radiance0(v3(0)), radiance1(v3(0)), pixelOffset(0), active(alive) {}
};

// RenderStructs.h:42-53 — identity rotation, origin, zNear -0.1, zFar -100, fov pi/2.
struct Camera : ptss_camera {
    Camera() {
        rotation = ptv::q4(1, 0, 0, 0);
        position = v3(0);
        zNear = -0.1f;
        zFar = -100.0f;
        fieldOfView = ptm::kPi / 2.0f;
    }
};

struct PointLight : ptss_point_light {  // RenderStructs.h:56-63
    PointLight(vec3 position_, vec3 power_) {
        position = position_;
        power = power_;
    }
};

struct AreaLight : ptss_area_light {  // RenderStructs.h:66-75
    AreaLight(vec3 power_, int triangleIdx_, size_t numTriangles_, float area_) {
        power = power_;
        area = area_;
        triangleIdx = triangleIdx_;
        numTriangles = numTriangles_;
    }
};

// RenderStructs.h:80-107. The reference leaves `roughness` (and, in the emitter constructor,
// `flags` by in-class default 0) uninitialised; SURVEY.md §9.4 DECISION: roughness = 0.
struct Material : ptss_material {
    Material() : ptss_material{} { indexOfRefraction = 1.0f; }
    Material(vec3 diffuseColor_, float diffAvg_, vec3 specularColor_ = v3(0), float specularExponent_ = 0,
             float specAvg_ = 0, float indexOfRefraction_ = 1.0f, vec3 absorption_ = v3(0), float refrAvg_ = 0,
             vec3 emmitance_ = v3(0))
        : ptss_material{} {
        diffuseColor = diffuseColor_;
        specularColor = specularColor_;
        absorption = absorption_;
        emmitance = emmitance_;
        specularExponent = specularExponent_;
        indexOfRefraction = indexOfRefraction_;
        diffAvg = diffAvg_;
        specAvg = specAvg_;
        refrAvg = refrAvg_;
    }
    explicit Material(vec3 emmitance_) : ptss_material{} {
        emmitance = emmitance_;
        indexOfRefraction = 1.0f;
    }
};

struct SurfaceElement {  // RenderStructs.h:110-121
    vec3 point;
    vec3 normal;
    int materialIdx;
};

static_assert(sizeof(Camera) == 40 && sizeof(PointLight) == 24 && sizeof(AreaLight) == 32 && sizeof(Material) == 76,
              "boundary layouts must match the reference (SURVEY.md §2.1)");
