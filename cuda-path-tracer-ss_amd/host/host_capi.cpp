// host_capi.cpp — extern "C" face of the host mirror (include/ptss_host.h). Host only; no HIP.
#include "ptss_host.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "HostOps.h"
#include "Scene.h"
#include "ptquant.h"
#include "pttri.h"
#include "xorwow.h"

struct ptss_scene {
    Scene scene;
};

// Triangle::intersectRay (Primitives.h:25-83) for n (triangle, ray) pairs in the general form and in the edge-class form
// the kernels pick for that triangle (csrc/pttri.h — the very functions the kernels call, compiled for the host).
namespace {
using namespace ptv;
template <int kC1, int kC2>
void triangleForm(const float* t, const float* o3, const float* d3, float limit, bool primary, float* out) {
    const vec3 v0 = v3(t[0], t[1], t[2]), e1 = v3(t[3], t[4], t[5]), e2 = v3(t[6], t[7], t[8]);
    const vec3 o = v3(o3[0], o3[1], o3[2]), d = v3(d3[0], d3[1], d3[2]);
    pttri::Head h;
    if (primary) {   // what primaryPrepKernel stores, with the general form
        const vec3 s = o - v0, r = cross(s, e1);
        h = pttri::head<kC1, kC2, true>(v0, e1, e2, s, r, dot(e2, r), o, d);
    } else {
        h = pttri::head<kC1, kC2, false>(v0, e1, e2, v3(0, 0, 0), v3(0, 0, 0), 0.0f, o, d);
    }
    float b0 = 0, b1 = 0, b2 = 0;
    const bool pass = pttri::passesHead(h, limit);
    if (pass) pttri::weights<kC1, kC2>(h, d, b0, b1, b2);
    out[0] = (pass && pttri::passesWeights(b0, b1, b2)) ? 1.0f : 0.0f;
    out[1] = h.dist;
    out[2] = b0;
    out[3] = b1;
    out[4] = b2;
    out[5] = h.det;
}
}  // namespace

extern "C" {

int ptss_scene_create(const char* preset, ptss_scene** out) {
    if (!preset || !out) return PTSS_HOST_EINVAL;
    ptss_scene* s = new (std::nothrow) ptss_scene();
    if (!s) return PTSS_HOST_EINVAL;
    if (!s->scene.buildPreset(preset)) {
        delete s;
        return PTSS_HOST_EINVAL;
    }
    *out = s;
    return PTSS_HOST_OK;
}

void ptss_scene_destroy(ptss_scene* s) { delete s; }

int ptss_scene_describe(const ptss_scene* s, ptss_scene_desc* out) {
    if (!s || !out) return PTSS_HOST_EINVAL;
    *out = s->scene.desc();
    return PTSS_HOST_OK;
}

int ptss_camera_default(ptss_camera* out) {
    if (!out) return PTSS_HOST_EINVAL;
    *out = Camera();
    return PTSS_HOST_OK;
}

int ptss_camera_move(ptss_camera* cam, unsigned char key, int* moved) {
    if (!cam) return PTSS_HOST_EINVAL;
    Camera c;
    static_cast<ptss_camera&>(c) = *cam;
    const bool m = moveCamera(c, key);
    *cam = c;
    if (moved) *moved = m ? 1 : 0;
    return PTSS_HOST_OK;
}

int ptss_write_tga(const char* filename, const ptss_uchar4* rgba, int width, int height) {
    if (!filename || !rgba || width <= 0 || height <= 0) return PTSS_HOST_EINVAL;
    return writeTga(filename, rgba, width, height) ? PTSS_HOST_OK : PTSS_HOST_EIO;
}

int ptss_tile_rows(int height, int band_rows, int rank, int world, int* rows, int cap) {
    if (height < 0 || band_rows <= 0 || world <= 0 || rank < 0 || rank >= world) return PTSS_HOST_EINVAL;
    int n = 0;
    for (int y = 0; y < height; ++y) {
        if ((y / band_rows) % world != rank) continue;
        if (rows && n < cap) rows[n] = y;
        ++n;
    }
    return n;
}

int ptss_probe_math(int op, const float* x, const float* y, float* out, size_t n) {
    if (!x || !out) return PTSS_HOST_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
            case 0: ptm::sincos(x[i], s, c); out[i] = s; break;
            case 1: ptm::sincos(x[i], s, c); out[i] = c; break;
            case 2: out[i] = ptm::tan(x[i]); break;
            case 3: out[i] = ptm::atan(x[i]); break;
            case 4: out[i] = ptm::log(x[i]); break;
            case 5: out[i] = ptm::exp(x[i]); break;
            case 6: if (!y) return PTSS_HOST_EINVAL; out[i] = ptm::pow(x[i], y[i]); break;
            case 7: out[i] = ptm::sqrt(x[i]); break;
            default: return PTSS_HOST_EINVAL;
        }
    }
    return PTSS_HOST_OK;
}

int ptss_probe_triangle_forms(const float* tri9, const float* o3, const float* d3, const float* limit, int primary, size_t n, int* cls,
                              float* general6, float* classed6) {
    if (!tri9 || !o3 || !d3 || !limit || !cls || !general6 || !classed6) return PTSS_HOST_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        const float* t = tri9 + 9 * i;
        const int c = pttri::triangleClass(v3(t[3], t[4], t[5]), v3(t[6], t[7], t[8]));
        cls[i] = c;
        triangleForm<0, 0>(t, o3 + 3 * i, d3 + 3 * i, limit[i], primary != 0, general6 + 6 * i);
        float* out = classed6 + 6 * i;
        switch (c) {
#define PTSS_TRI_CASE(c1, c2) \
    case (c1) * 4 + (c2): triangleForm<c1, c2>(t, o3 + 3 * i, d3 + 3 * i, limit[i], primary != 0, out); break;
            PTSS_TRI_CASE(0, 1) PTSS_TRI_CASE(0, 2) PTSS_TRI_CASE(0, 3)
            PTSS_TRI_CASE(1, 0) PTSS_TRI_CASE(1, 2) PTSS_TRI_CASE(1, 3)
            PTSS_TRI_CASE(2, 0) PTSS_TRI_CASE(2, 1) PTSS_TRI_CASE(2, 3)
            PTSS_TRI_CASE(3, 0) PTSS_TRI_CASE(3, 1) PTSS_TRI_CASE(3, 2)
#undef PTSS_TRI_CASE
            default: triangleForm<0, 0>(t, o3 + 3 * i, d3 + 3 * i, limit[i], primary != 0, out); break;
        }
    }
    return PTSS_HOST_OK;
}

int ptss_probe_quantize(const float* x, unsigned int* out, size_t n) {
    if (!x || !out) return PTSS_HOST_EINVAL;
    for (size_t i = 0; i < n; ++i) out[i] = ptq::quantize_literal(x[i]);
    return PTSS_HOST_OK;
}

int ptss_probe_quant_table(float* out257) {
    if (!out257) return PTSS_HOST_EINVAL;
    float T[ptq::kTableFloats];
    if (!ptq::build_thresholds(T)) return PTSS_HOST_EINVAL;
    for (int k = 0; k <= 256; ++k) out257[k] = T[k];
    return PTSS_HOST_OK;
}

static const uint32_t* jumpTable() {
    static std::vector<uint32_t> table;
    if (table.empty()) {
        table.resize(ptrng::kJumpTableWords);
        ptrng::build_subsequence_table(table.data());
    }
    return table.data();
}

int ptss_probe_rng_init(unsigned long long seed, unsigned int subsequence, unsigned int* out6) {
    if (!out6) return PTSS_HOST_EINVAL;
    ptrng::State s = ptrng::seeded(seed);
    ptrng::skip_subsequences(s, subsequence, jumpTable());
    for (int i = 0; i < 5; ++i) out6[i] = s.v[i];
    out6[5] = s.d;
    return PTSS_HOST_OK;
}

int ptss_probe_rng_draw(unsigned int* state6, unsigned int* raw, float* uni, size_t n) {
    if (!state6) return PTSS_HOST_EINVAL;
    ptrng::State s;
    for (int i = 0; i < 5; ++i) s.v[i] = state6[i];
    s.d = state6[5];
    for (size_t i = 0; i < n; ++i) {
        ptrng::State before = s;
        const uint32_t r = ptrng::next(s);
        if (raw) raw[i] = r;
        if (uni) uni[i] = ptrng::uniform(before);
    }
    for (int i = 0; i < 5; ++i) state6[i] = s.v[i];
    state6[5] = s.d;
    return PTSS_HOST_OK;
}

int ptss_probe_rng_jump_table(unsigned int* out, size_t words) {
    if (!out || words != (size_t)ptrng::kJumpTableWords) return PTSS_HOST_EINVAL;
    std::memcpy(out, jumpTable(), words * sizeof(uint32_t));
    return PTSS_HOST_OK;
}

}  // extern "C"
