// Scene.cpp — procedural scenes with the reference's content (CudaTracer/Scene.cpp:17-371),
// written table-first: every box is a list of {placement, material slot} rows fed to one helper.
// Numbers (colours, exponents, IORs, wall angles, light powers) are the reference's data and are
// cited per block; the code that assembles them is this repo's own.
#include "Scene.h"

namespace {

const vec3 kAxisX = {1, 0, 0};
const vec3 kAxisY = {0, 1, 0};

struct WallRow {
    vec3 shift;     // translate(...)
    float degrees;  // rotate(degrees, axis); 0 = no rotation factor
    vec3 axis;
    float edge;     // uniform scale
    int slot;       // material slot relative to the box's first material
};

mat4 place(const WallRow& w) {
    mat4 m = translate(w.shift);
    if (w.degrees != 0.0f) m = m * rotate(w.degrees, w.axis);
    return m * scale(v3(w.edge));
}

// glass: white diffuse colour with zero diffuse weight, white mirror-like specular (exponent inf),
// specAvg/refrAvg 0.7, IOR 1.55, coloured absorption           (reference Scene.cpp:128-140)
Material glass(vec3 absorption) {
    return Material(v3(1.0f), 0.0f, v3(1.0f), PTSS_INFINITY, 0.7f, 1.55f, absorption, 0.7f);
}

// Cook-Torrance lobe: diffAvg .1, specAvg .6, IOR 1.7, exponent inf (reference Scene.cpp:198-210)
Material cookTorrance(vec3 diffuse, vec3 specular, float roughness) {
    Material m(diffuse, 0.1f, specular, PTSS_INFINITY, 0.6f, 1.7f);
    m.flags |= MAT_FLAG_COOK_TORRANCE;
    m.roughness = roughness;
    return m;
}

}  // namespace

Scene::Scene() {}
Scene::~Scene() {}

// rand() of the reference is MSVC v120's, unseeded (Scene.cpp:3, CudaTracer.vcxproj): the LCG below with RAND_MAX =
// 32767 [unverifiable offline]. UNPINNED ASSUMPTION beside it: the reference draws a sphere's coordinates as constructor
// ARGUMENTS — vec3(rnd(5.0f) - 2.5f, rnd(5.0f) - 2.5f, rnd(7.0f) - 9.0f) (Scene.cpp:161, 219) — and C++ leaves the
// evaluation order of function arguments unspecified (MSVC commonly evaluates them right to left). The generators here
// draw x, then y, then z (left to right), so the "same spheres every time" layout of the default / mixed / lambert
// presets is self-consistent but may be the reference's with x and z draws swapped (`positionDrawsRightToLeft`, preset
// suffix "@rtl", builds that other layout); nothing the reference holds can decide it (its image.tga shows the defined spheres of the Cornell preset, which draw nothing). Statistically the two
// layouts are the same scene: 20 spheres of the same size distribution in the same box.
int Scene::nextRand() {
    randState = randState * 214013u + 2531011u;
    return (int)((randState >> 16) & 0x7fffu);
}
// `#define rnd(x) (x * rand() / RAND_MAX)` with a float x (Scene.cpp:3): float*int, then /int.
float Scene::rnd(float x) { return x * (float)nextRand() / (float)32767; }
// `rnd(1);` statements (Scene.cpp:159,217) are integer expressions whose only effect is a draw.
void Scene::burn(int draws) {
    for (int i = 0; i < draws; ++i) (void)nextRand();
}

// Scene.cpp:17-32 — the code's default: 5 Cook-Torrance + 15 glass spheres in a 10-unit mirror box.
void Scene::build() {
    addRandomSpheres(5);
    addRandomGlassSpheres(15);
    addMirrorBox(10);
}

// Scene.cpp:63-96 — unit square in z=0 pushed through `transformation`; two triangles
// (v0,v1,v2),(v3,v1,v2) sharing one flat normal = normalize(inverse(transpose(M)) * (0,0,1,0)).
void Scene::addRectangularModel(mat4 transformation, int materialIdx) {
    vec3 corner[4];
    for (int k = 0; k < 4; ++k) {
        const vec4 p = transformation * vec4{(float)(k >> 1) - 0.5f, (float)(k & 1) - 0.5f, 0.0f, 1.0f};
        corner[k] = v3(p.x, p.y, p.z);
    }
    const vec4 n4 = inverse(transpose(transformation)) * vec4{0.0f, 0.0f, 1.0f, 0.0f};
    const vec3 n = ptv::normalize(v3(n4.x, n4.y, n4.z));
    trianglesVec.push_back(Triangle(corner[0], corner[1], corner[2], n, n, n, materialIdx));
    trianglesVec.push_back(Triangle(corner[3], corner[1], corner[2], n, n, n, materialIdx));
}

// Scene.cpp:40-55 — rectangle + AreaLight{power, first triangle index, 2, |e1 x e2|}.
void Scene::addAreaLight(mat4 transformation, int materialIdx, vec3 power) {
    using namespace ptv;
    const size_t first = trianglesVec.size();
    addRectangularModel(transformation, materialIdx);
    const Triangle& t = trianglesVec[first];
    const vec3 e1 = t.vertex1 - t.vertex2;
    const vec3 e2 = t.vertex2 - t.vertex0;
    areaLightsVec.push_back(AreaLight(power, (int)first, 2, length(cross(e1, e2))));
}

// Scene.cpp:98-109 — one Phong-glass (exponent 300) and one red Phong (exponent 250) sphere.
void Scene::addDefinedSpheres(const float size) {
    const int first = (int)materialsVec.size();
    materialsVec.push_back(Material(v3(1.0f, 0.0f, 0.0f), 0.35f, v3(1.0f), 250, 0.6f, 2.5f));
    materialsVec.push_back(Material(v3(1.0f), 0.0f, v3(1.0f), 300, 0.9f, 1.55f, v3(0.15f, 0.15f, 0.0f), 0.9f));
    spheresVec.push_back(Sphere(v3(-2, -(size - 1.5f), -(size * 1.3f)), 1.5f, first + 1));
    spheresVec.push_back(Sphere(v3(1, -(size - 1.0f), -(size * 1.4f)), 1.0f, first));
}

// Scene.cpp:115-167 — three absorbing glasses; per sphere 3 burnt draws, then x,y,z,radius.
void Scene::addRandomGlassSpheres(const size_t numSpheres) {
    const int first = (int)materialsVec.size();
    materialsVec.push_back(glass(v3(0.0f, 0.75f, 0.75f)));  // "red glass" (absorbs G,B)
    materialsVec.push_back(glass(v3(0.75f, 0.75f, 0.0f)));  // "blue glass"
    materialsVec.push_back(glass(v3(0.75f, 0.0f, 0.75f)));  // "green glass"
    for (size_t i = 0; i < numSpheres; ++i) {
        burn(3);
        float x, y, z;
        if (positionDrawsRightToLeft) {  // the other legal order of vec3(...)'s arguments (see nextRand)
            z = rnd(7.0f) - 9.0f;
            y = rnd(5.0f) - 2.5f;
            x = rnd(5.0f) - 2.5f;
        } else {
            x = rnd(5.0f) - 2.5f;
            y = rnd(5.0f) - 2.5f;
            z = rnd(7.0f) - 9.0f;
        }
        const float r = rnd(1.0f) + 0.2f;
        spheresVec.push_back(Sphere(v3(x, y, z), r, first + (int)(i % 3)));
    }
}

// Scene.cpp:173-225 — three Cook-Torrance materials; per sphere 2 burnt draws, then x,y,z,radius.
void Scene::addRandomSpheres(const size_t numSpheres) {
    const int first = (int)materialsVec.size();
    materialsVec.push_back(cookTorrance(v3(1.0f, 0.1f, 0.1f), v3(1.0f, 0.2f, 0.2f), 0.3f));
    materialsVec.push_back(cookTorrance(v3(0.1f, 0.1f, 1.0f), v3(0.2f, 0.2f, 1.0f), 0.1f));
    materialsVec.push_back(cookTorrance(v3(0.1f, 1.0f, 0.1f), v3(0.2f, 1.0f, 0.2f), 0.5f));
    for (size_t i = 0; i < numSpheres; ++i) {
        burn(2);
        float x, y, z;
        if (positionDrawsRightToLeft) {  // the other legal order of vec3(...)'s arguments (see nextRand)
            z = rnd(7.0f) - 9.0f;
            y = rnd(5.0f) - 2.5f;
            x = rnd(5.0f) - 2.5f;
        } else {
            x = rnd(5.0f) - 2.5f;
            y = rnd(5.0f) - 2.5f;
            z = rnd(7.0f) - 9.0f;
        }
        const float r = rnd(1.0f) + 0.2f;
        spheresVec.push_back(Sphere(v3(x, y, z), r, first + (int)(i % 3)));
    }
}

// Scene.cpp:231-294 — open-front box: white floor/ceiling/back, red left, green right, a mirror
// panel (specAvg .8, IOR 5.8, no PURE_REFLECTION flag) just inside the right wall, 2.5^2 ceiling light 400.
void Scene::addCornellBox(const float wallSize) {
    const int first = (int)materialsVec.size();
    materialsVec.push_back(Material(v3(1.0f, 1.0f, 0.8f), 0.7f));                                   // +0 white
    materialsVec.push_back(Material(v3(1.0f, 0.0f, 0.0f), 0.7f));                                   // +1 red
    materialsVec.push_back(Material(v3(0.0f, 1.0f, 0.0f), 0.7f));                                   // +2 green
    materialsVec.push_back(Material(v3(1.0f, 1.0f, 1.0f)));                                         // +3 emitter
    materialsVec.push_back(Material(v3(0.0f), 0.0f, v3(1.0f), PTSS_INFINITY, 0.8f, 5.8f));          // +4 mirror

    const float h = wallSize / 2;
    const WallRow walls[] = {
        {v3(0, -h, -h), -90.0f, kAxisX, wallSize, 0},              // floor
        {v3(0, h, -h), 90.0f, kAxisX, wallSize, 0},                // ceiling
        {v3(-h, 0, -h), 90.0f, kAxisY, wallSize, 1},               // left
        {v3(h, 0, -h), -90.0f, kAxisY, wallSize, 2},               // right
        {v3(h - 0.02f, 0, -h), -90.0f, kAxisY, wallSize - 2, 4},   // mirror panel
        {v3(0, 0, -wallSize), 0.0f, kAxisY, wallSize, 0},          // back
    };
    for (const WallRow& w : walls) addRectangularModel(place(w), first + w.slot);

    const float power = 400;
    addAreaLight(place({v3(0, h - 0.01f, -h), 90.0f, kAxisX, 2.5f, 3}), first + 3, v3(power));
}

// Scene.cpp:301-371 — closed box: white floor/back/front, PURE_REFLECTION mirror (specAvg .9,
// IOR 5.8) on ceiling, right and an 88-degree left wall pulled in by 0.2*offset; ceiling light
// (100,400,400) 2.5^2 and a floor light (133.3,0,133.3) 1.5^2 with a violet emitter material.
void Scene::addMirrorBox(const float wallSize) {
    const int first = (int)materialsVec.size();
    materialsVec.push_back(Material(v3(1.0f, 1.0f, 0.8f), 0.7f));                                   // +0 white
    materialsVec.push_back(Material(v3(1.0f, 0.0f, 0.0f), 0.7f));                                   // +1 red (unused)
    materialsVec.push_back(Material(v3(0.0f, 1.0f, 0.0f), 0.7f));                                   // +2 green (unused)
    materialsVec.push_back(Material(v3(1.0f, 1.0f, 1.0f)));                                         // +3 white emitter
    Material mirror(v3(0.0f), 0.0f, v3(1.0f), PTSS_INFINITY, 0.9f, 5.8f);                           // +4 mirror
    mirror.flags |= MAT_FLAG_PURE_REFLECTION;
    materialsVec.push_back(mirror);
    materialsVec.push_back(Material(v3(1.0f, 0.6f, 1.0f)));                                         // +5 violet emitter

    const float h = wallSize / 2;
    // `-offset + .2 * offset` is evaluated in double in the reference and narrowed by vec3's ctor
    const float leftX = (float)(-(double)h + 0.2 * (double)h);
    const WallRow walls[] = {
        {v3(0, -h, -h), -90.0f, kAxisX, wallSize, 0},      // floor
        {v3(0, h, -h), 90.0f, kAxisX, wallSize, 4},        // ceiling (mirror)
        {v3(leftX, 0, -h), 88.0f, kAxisY, wallSize, 4},    // left (mirror, 88 degrees)
        {v3(h, 0, -h), -90.0f, kAxisY, wallSize, 4},       // right (mirror)
        {v3(0, 0, -wallSize), 0.0f, kAxisY, wallSize, 0},  // back
        {v3(0, 0, 0), 180.0f, kAxisY, wallSize, 0},        // front, through the camera plane
    };
    for (const WallRow& w : walls) addRectangularModel(place(w), first + w.slot);

    const float power = 400;
    addAreaLight(place({v3(0, h - 0.01f, -h), 90.0f, kAxisX, 2.5f, 3}), first + 3, v3(power / 4, power, power));
    addAreaLight(place({v3(0, -h + 0.01f, -h), -90.0f, kAxisX, 1.5f, 5}), first + 5, v3(power / 3, 0, power / 3));
}

// ------------------------------------------------------------------------------------------------
// Additions (not in the reference): presets for BASELINE.json's five configs, SURVEY.md §9.6.
// ------------------------------------------------------------------------------------------------

// n spheres, uniformly placed in [-halfXY,halfXY]^2 x [zFar,zNear], radius in [rMin,rMax], cycling
// through every material present when called. Same draw discipline as addRandomSpheres.
void Scene::addSphereField(size_t numSpheres, float halfXY, float zNear, float zFar, float rMin, float rMax) {
    const int numMaterials = (int)materialsVec.size();
    for (size_t i = 0; i < numSpheres; ++i) {
        burn(2);
        const float x = rnd(2 * halfXY) - halfXY;
        const float y = rnd(2 * halfXY) - halfXY;
        const float z = rnd(zNear - zFar) + zFar;
        const float r = rnd(rMax - rMin) + rMin;
        spheresVec.push_back(Sphere(v3(x, y, z), r, numMaterials ? (int)(i % (size_t)numMaterials) : 0));
    }
}

// C2: every non-emissive material becomes Material(colour, 0.7f) — pure Lambert.
void Scene::makeLambertOnly() {
    for (Material& m : materialsVec) {
        const bool emits = m.emmitance.x > 0 || m.emmitance.y > 0 || m.emmitance.z > 0;
        if (emits) continue;
        const bool hasDiffuse = m.diffuseColor.x > 0 || m.diffuseColor.y > 0 || m.diffuseColor.z > 0;
        m = Material(hasDiffuse ? m.diffuseColor : m.specularColor, 0.7f);
    }
}

bool Scene::buildPreset(const std::string& preset) {
    // "<name>@rtl": the reference's random-sphere generators with their position arguments drawn right to left
    std::string name = preset;
    positionDrawsRightToLeft = false;   // a property of THIS build, not of the object: a later preset without the suffix draws left to right again
    const size_t at = name.find("@rtl");
    if (at != std::string::npos && at + 4 == name.size()) {
        positionDrawsRightToLeft = true;
        name.erase(at);
    }
    if (name == "default") {  // C1 literal default, also the geometry of C2
        build();
    } else if (name == "cornell") {  // C1 as BASELINE words it; matches CudaTracer/image.tga
        addDefinedSpheres(4);
        addCornellBox(8);
    } else if (name == "lambert") {  // C2
        build();
        makeLambertOnly();
    } else if (name == "mixed") {  // C3/C4: default + the two finite-exponent Phong spheres
        addRandomSpheres(5);
        addRandomGlassSpheres(15);
        addDefinedSpheres(4);
        addMirrorBox(10);
    } else if (name == "stress") {  // C5: 1,024 spheres of every material class in an open box
        materialsVec.push_back(cookTorrance(v3(1.0f, 0.1f, 0.1f), v3(1.0f, 0.2f, 0.2f), 0.3f));
        materialsVec.push_back(glass(v3(0.0f, 0.75f, 0.75f)));
        materialsVec.push_back(Material(v3(0.2f, 0.4f, 1.0f), 0.7f));
        materialsVec.push_back(Material(v3(1.0f, 0.0f, 0.0f), 0.35f, v3(1.0f), 250, 0.6f, 2.5f));
        addSphereField(1024, 3.6f, -1.5f, -7.7f, 0.05f, 0.22f);
        addCornellBox(8);
    } else if (name == "pointlight") {  // §8f-3: the commented-out point lights of Scene.cpp:21-22
        addDefinedSpheres(4);
        addCornellBox(8);
        areaLightsVec.clear();
        const float power = 500;
        pointLightsVec.push_back(PointLight(v3(0, 0.0f, 2.5f), v3(power)));
        pointLightsVec.push_back(PointLight(v3(2, 3.0f, -5), v3(power)));
    } else {
        return false;
    }
    return true;
}

ptss_scene_desc Scene::desc(vec3 defaultColor) const {
    ptss_scene_desc d{};
    d.spheres = spheresVec.data();
    d.numSpheres = spheresVec.size();
    d.triangles = trianglesVec.data();
    d.numTriangles = trianglesVec.size();
    d.materials = materialsVec.data();
    d.numMaterials = materialsVec.size();
    d.pointLights = pointLightsVec.data();
    d.numPointLights = pointLightsVec.size();
    d.areaLights = areaLightsVec.data();
    d.numAreaLights = areaLightsVec.size();
    d.defaultColor = defaultColor;
    return d;
}
