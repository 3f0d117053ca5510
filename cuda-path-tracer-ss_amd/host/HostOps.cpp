#include "HostOps.h"

#include <cstdio>
#include <vector>

using namespace ptv;

quat quatFromEuler(vec3 e) {
    float sx, cx, sy, cy, sz, cz;
    ptm::sincos(e.x * 0.5f, sx, cx);
    ptm::sincos(e.y * 0.5f, sy, cy);
    ptm::sincos(e.z * 0.5f, sz, cz);
    return q4(cx * cy * cz + sx * sy * sz,
              sx * cy * cz - cx * sy * sz,
              cx * sy * cz + sx * cy * sz,
              cx * cy * sz - sx * sy * cz);
}

bool moveCamera(Camera& camera, unsigned char key) {
    const float step = 0.2f;
    const float turn = 10.0f * ptm::kPi / 180.0f;
    struct Move { unsigned char key; vec3 delta; };
    static const Move moves[] = {
        {'w', {0, 0, -step}}, {'a', {-step, 0, 0}}, {'s', {0, 0, step}},
        {'d', {step, 0, 0}},  {'q', {0, step, 0}},  {'e', {0, -step, 0}},
    };
    for (const Move& m : moves)
        if (m.key == key) {
            camera.position = camera.position + rotate(camera.rotation, m.delta);
            return true;
        }
    struct Turn { unsigned char key; vec3 euler; };
    const Turn turns[] = {
        {'f', {0, turn, 0}}, {'h', {0, -turn, 0}}, {'g', {-turn, 0, 0}}, {'t', {turn, 0, 0}},
    };
    for (const Turn& t : turns)
        if (t.key == key) {
            camera.rotation = normalize(mul(camera.rotation, quatFromEuler(t.euler)));
            return true;
        }
    return false;
}

bool writeTga(const char* filename, const ptss_uchar4* rgba, int width, int height) {
    std::FILE* f = std::fopen(filename, "wb");
    if (!f) return false;
    unsigned char header[18] = {0};
    header[2] = 2;  // uncompressed true-colour
    header[12] = (unsigned char)(width % 256);
    header[13] = (unsigned char)(width / 256);
    header[14] = (unsigned char)(height % 256);
    header[15] = (unsigned char)(height / 256);
    header[16] = 24;
    std::vector<unsigned char> body((size_t)width * height * 3);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        body[3 * i + 0] = rgba[i].z;
        body[3 * i + 1] = rgba[i].y;
        body[3 * i + 2] = rgba[i].x;
    }
    bool ok = std::fwrite(header, 1, 18, f) == 18 && std::fwrite(body.data(), 1, body.size(), f) == body.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}
