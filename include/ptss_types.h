/* ptss_types.h — plain-C layouts of the scene/camera records that cross the drop-in boundary.
 *
 * Field order and sizes mirror the reference's host/device-shared structs so a maintainer can
 * hand the reference's std::vector<T>::data() straight to ptss_create():
 *   ptss_sphere      <- class Sphere      CudaTracer/Primitives.h:86-93       (20 B)
 *   ptss_triangle    <- class Triangle    CudaTracer/Primitives.h:6-16        (76 B)
 *   ptss_material    <- struct Material   CudaTracer/RenderStructs.h:80-107   (76 B, flags at 72)
 *   ptss_point_light <- struct PointLight CudaTracer/RenderStructs.h:56-63    (24 B)
 *   ptss_area_light  <- struct AreaLight  CudaTracer/RenderStructs.h:66-75    (32 B)
 *   ptss_camera      <- struct Camera     CudaTracer/RenderStructs.h:42-53    (40 B)
 *   ptss_uchar4      <- CUDA uchar4 (display pixel, RGBA)  CudaTracer/CudaTracer.cu:88-101
 * glm::vec3 is three packed floats; glm::quat is stored x,y,z,w (its constructor takes w first).
 */
#ifndef PTSS_TYPES_H
#define PTSS_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptss_vec3 { float x, y, z; } ptss_vec3;
typedef struct ptss_quat { float x, y, z, w; } ptss_quat;
typedef struct ptss_uchar4 { unsigned char x, y, z, w; } ptss_uchar4;

typedef struct ptss_sphere {
    ptss_vec3 position;
    float radius;
    int materialIdx;
} ptss_sphere;

typedef struct ptss_triangle {
    ptss_vec3 vertex0, vertex1, vertex2;
    ptss_vec3 normal0, normal1, normal2;
    int materialIdx;
} ptss_triangle;

#define PTSS_MAT_FLAG_PURE_REFLECTION 0x01 /* RenderStructs.h:77 */
#define PTSS_MAT_FLAG_COOK_TORRANCE 0x03   /* RenderStructs.h:78 (overlaps bit 0 — kept literal) */

typedef struct ptss_material {
    ptss_vec3 diffuseColor;
    ptss_vec3 specularColor;
    ptss_vec3 absorption;
    ptss_vec3 emmitance; /* sic — reference spelling */
    float specularExponent;
    float indexOfRefraction;
    float diffAvg;
    float specAvg;
    float refrAvg;
    float roughness;
    char flags;
} ptss_material;

typedef struct ptss_point_light {
    ptss_vec3 position;
    ptss_vec3 power;
} ptss_point_light;

typedef struct ptss_area_light {
    ptss_vec3 power;
    float area;
    int triangleIdx;
    size_t numTriangles;
} ptss_area_light;

typedef struct ptss_camera {
    ptss_quat rotation;
    ptss_vec3 position;
    float zNear;
    float zFar;
    float fieldOfView;
} ptss_camera;

/* The five scene vectors of class Scene (CudaTracer/Scene.h:11-15), as uploaded verbatim by
 * main (CudaTracer/CudaTracer.cu:696-700), plus RendererData::defaultColor (CudaTracer.h:15). */
typedef struct ptss_scene_desc {
    const ptss_sphere* spheres;
    size_t numSpheres;
    const ptss_triangle* triangles;
    size_t numTriangles;
    const ptss_material* materials;
    size_t numMaterials;
    const ptss_point_light* pointLights;
    size_t numPointLights;
    const ptss_area_light* areaLights;
    size_t numAreaLights;
    ptss_vec3 defaultColor;
} ptss_scene_desc;

#ifdef __cplusplus
}
#endif
#endif
