// Scene.h — the reference's scene container and builder API (CudaTracer/Scene.h:5-27), kept name
// for name so the host loop above the C-ABI reads like the reference's main():
//   five public std::vectors, addRectangularModel / addAreaLight / build / addRandomSpheres /
//   addRandomGlassSpheres / addCornellBox / addMirrorBox / addDefinedSpheres.
// Additions for BASELINE.json's configs: buildPreset() and the helpers it uses.
#pragma once
#include <string>
#include <vector>
#include "Mat4.h"
#include "Primitives.h"

class Scene {
public:
    Scene();
    ~Scene();

    std::vector<Sphere> spheresVec;
    std::vector<Triangle> trianglesVec;
    std::vector<Material> materialsVec;
    std::vector<PointLight> pointLightsVec;
    std::vector<AreaLight> areaLightsVec;

    void addRectangularModel(mat4 transformation, int materialIdx);
    void addAreaLight(mat4 transformation, int materialIdx, vec3 power);

    void build();
    void addRandomSpheres(const size_t numSpheres);
    void addRandomGlassSpheres(const size_t numSpheres);
    void addCornellBox(const float wallSize);
    void addMirrorBox(const float wallSize);
    void addDefinedSpheres(const float size);

    // --- additions -------------------------------------------------------------------------
    // "default" | "cornell" | "lambert" | "mixed" | "stress" | "pointlight"; false if unknown.
    // A "@rtl" suffix (e.g. "default@rtl") draws the random spheres' position arguments right to left (z, y, x): the order
    // C++ leaves unspecified at Scene.cpp:161, 219 of the reference and MSVC commonly takes; default: left to right.
    bool buildPreset(const std::string& name);
    bool positionDrawsRightToLeft = false;
    void addSphereField(size_t numSpheres, float halfXY, float zNear, float zFar, float rMin, float rMax);
    void makeLambertOnly();
    ptss_scene_desc desc(vec3 defaultColor = v3(0)) const;

private:
    // libc rand() stand-in: MSVC's LCG, unseeded (state 1), RAND_MAX 32767 (SURVEY.md §9.5 DECISION).
    unsigned int randState = 1u;
    int nextRand();
    float rnd(float x);
    void burn(int draws);
};
