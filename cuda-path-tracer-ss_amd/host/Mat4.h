// Mat4.h — the small column-major 4x4 toolkit the scene builder needs in place of glm
// (reference call sites: glm::translate/scale/rotate/inverse/transpose in CudaTracer/Scene.cpp:51,83,
// 250-370). rotate() takes DEGREES, like glm 0.9.5.4 without GLM_FORCE_RADIANS.
// Trig goes through ptm::sincos so the primitive table is identical on every host.
#pragma once
#include "ptmath.h"

struct vec4 {
    float x, y, z, w;
};

struct mat4 {
    float c[4][4];  // c[column][row]
    static mat4 identity() {
        mat4 m{};
        for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f;
        return m;
    }
};

inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r{};
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            float acc = a.c[0][row] * b.c[col][0];
            for (int k = 1; k < 4; ++k) acc = acc + a.c[k][row] * b.c[col][k];
            r.c[col][row] = acc;
        }
    return r;
}

inline vec4 operator*(const mat4& m, const vec4& v) {
    float in[4] = {v.x, v.y, v.z, v.w};
    float out[4];
    for (int row = 0; row < 4; ++row)
        out[row] = (m.c[0][row] * in[0] + m.c[1][row] * in[1]) + (m.c[2][row] * in[2] + m.c[3][row] * in[3]);
    return vec4{out[0], out[1], out[2], out[3]};
}

inline mat4 translate(ptv::vec3 t) {
    mat4 m = mat4::identity();
    m.c[3][0] = t.x; m.c[3][1] = t.y; m.c[3][2] = t.z;
    return m;
}

inline mat4 scale(ptv::vec3 s) {
    mat4 m = mat4::identity();
    m.c[0][0] = s.x; m.c[1][1] = s.y; m.c[2][2] = s.z;
    return m;
}

// Rodrigues rotation about `axis` by `degrees`.
inline mat4 rotate(float degrees, ptv::vec3 axis) {
    const float radians = degrees * (ptm::kPi / 180.0f);
    float s, c;
    ptm::sincos(radians, s, c);
    const ptv::vec3 a = ptv::normalize(axis);
    const float k = 1.0f - c;
    const float ax[3] = {a.x, a.y, a.z};
    // antisymmetric part: s * [a]x
    const float skew[3][3] = {{0, -a.z, a.y}, {a.z, 0, -a.x}, {-a.y, a.x, 0}};
    mat4 m = mat4::identity();
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) {
            float v = k * ax[row] * ax[col] + s * skew[row][col];
            if (row == col) v = c + k * ax[row] * ax[col];
            m.c[col][row] = v;
        }
    return m;
}

inline mat4 transpose(const mat4& a) {
    mat4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.c[i][j] = a.c[j][i];
    return r;
}

// General inverse by cofactors (adjugate / determinant).
inline mat4 inverse(const mat4& a) {
    auto at = [&](int row, int col) { return a.c[col][row]; };
    auto minor3 = [&](int skipRow, int skipCol) {
        int r[3], c[3], ri = 0, ci = 0;
        for (int i = 0; i < 4; ++i) {
            if (i != skipRow) r[ri++] = i;
            if (i != skipCol) c[ci++] = i;
        }
        return at(r[0], c[0]) * (at(r[1], c[1]) * at(r[2], c[2]) - at(r[1], c[2]) * at(r[2], c[1])) -
               at(r[0], c[1]) * (at(r[1], c[0]) * at(r[2], c[2]) - at(r[1], c[2]) * at(r[2], c[0])) +
               at(r[0], c[2]) * (at(r[1], c[0]) * at(r[2], c[1]) - at(r[1], c[1]) * at(r[2], c[0]));
    };
    float cof[4][4];
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) cof[row][col] = (((row + col) & 1) ? -1.0f : 1.0f) * minor3(row, col);
    float det = 0.0f;
    for (int col = 0; col < 4; ++col) det = det + at(0, col) * cof[0][col];
    const float inv = 1.0f / det;
    mat4 r{};
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) r.c[col][row] = cof[col][row] * inv;  // adjugate = cof^T
    return r;
}
