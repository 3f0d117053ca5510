// Primitives.h — host faces of the two primitive records (reference: CudaTracer/Primitives.h:6-23
// Triangle, :86-96 Sphere). Only the data + constructors live on the host; the intersectors
// (Primitives.h:25-83, :107-175) are device code in csrc/ptss_kernels.hip.
#pragma once
#include "RenderStructs.h"

class Triangle : public ptss_triangle {
public:
    Triangle() : ptss_triangle{} {}
    Triangle(vec3 v0, vec3 v1, vec3 v2, vec3 n0, vec3 n1, vec3 n2, int materialIdx_) {
        vertex0 = v0; vertex1 = v1; vertex2 = v2;
        normal0 = n0; normal1 = n1; normal2 = n2;
        materialIdx = materialIdx_;
    }
};

class Sphere : public ptss_sphere {
public:
    Sphere() : ptss_sphere{} {}
    Sphere(vec3 position_, float radius_, int materialIdx_) {
        position = position_;
        radius = radius_;
        materialIdx = materialIdx_;
    }
};

static_assert(sizeof(Triangle) == 76 && sizeof(Sphere) == 20, "boundary layouts must match the reference");
