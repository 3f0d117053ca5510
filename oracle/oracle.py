"""oracle.py — ctypes binding of oracle/_build/liboracle.so (oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product (cuda-path-tracer-ss_amd/) never imports this module.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "cuda-path-tracer-ss_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from ptss_types import Camera, SceneDesc, Sphere, Triangle  # noqa: E402  (layouts only, no product code)

LIB = os.path.join(_HERE, "_build", "liboracle.so")
LIB_LIBM = os.path.join(_HERE, "_build", "liboracle_libm.so")  # the same source against libm (oracle/libm_math.h)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_libs = {}


def lib(path=None):
    path = path or LIB
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python __graft_entry__.py build`")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int, C.c_ulonglong, C.c_uint, C.c_int, C.c_int]
        L.oracle_create.restype = vp
        L.oracle_destroy.argtypes = [vp]
        L.oracle_destroy.restype = None
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = None
        L.oracle_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.oracle_set_camera.restype = None
        L.oracle_set_mode.argtypes = [vp, C.c_int]
        L.oracle_set_mode.restype = None
        L.oracle_set_max_iterations.argtypes = [vp, C.c_uint]
        L.oracle_set_max_iterations.restype = None
        L.oracle_request_reset.argtypes = [vp]
        L.oracle_request_reset.restype = None
        L.oracle_generate_frame.argtypes = [vp, vp, C.c_int]
        L.oracle_generate_frame.restype = None
        L.oracle_accumulator.argtypes = [vp]
        L.oracle_accumulator.restype = _u32p
        L.oracle_float_sum.argtypes = [vp]
        L.oracle_float_sum.restype = _f32p
        L.oracle_last_radiance0.argtypes = [vp]
        L.oracle_last_radiance0.restype = _f32p
        L.oracle_live_counts.argtypes = [vp, _u32p, C.c_int]
        L.oracle_total_ray_bounces.argtypes = [vp]
        L.oracle_total_ray_bounces.restype = C.c_ulonglong
        L.oracle_rng_state.argtypes = [vp, C.c_long, C.c_int, _u32p]
        L.oracle_rng_state.restype = None
        L.oracle_probe_sphere.argtypes = [C.POINTER(Sphere), _f32p, C.c_float, _f32p]
        L.oracle_probe_triangle.argtypes = [C.POINTER(Triangle), _f32p, C.c_float, _f32p]
        L.oracle_probe_fresnel.argtypes = [C.c_float, C.c_float]
        L.oracle_probe_fresnel.restype = C.c_float
        L.oracle_probe_rotate_y_to.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_probe_rotate_y_to.restype = None
        L.oracle_probe_sampler.argtypes = [C.c_int, _f32p, C.c_float, C.c_ulonglong, C.c_int, _f32p]
        L.oracle_probe_sampler.restype = None
        L.oracle_probe_shade.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_ulonglong, _f32p]
        L.oracle_probe_shade.restype = None
        L.oracle_probe_quantize.argtypes = [C.c_float]
        L.oracle_probe_quantize.restype = C.c_uint
        L.oracle_probe_rng.argtypes = [C.c_ulonglong, C.c_uint, C.c_int, _u32p, _u32p, _f32p]
        L.oracle_probe_rng.restype = None
        L.oracle_probe_eye_ray.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Camera), C.c_ulonglong, _f32p]
        L.oracle_probe_eye_ray.restype = None
        _libs[path] = L
    return _libs[path]


def set_threads(n):
    lib().oracle_set_threads(int(n))


def max_threads():
    return int(lib().oracle_max_threads())


def cpu_share():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Oracle:
    """CPU generateFrame over a full frame. scene_desc: a ptss_types.SceneDesc (kept alive by the caller)."""

    def __init__(self, scene_desc, width, height, max_iterations=15, seed=0x5EED, literal_slot_rng=False,
                 samples_per_pass=1, math="ptmath"):
        """math="ptmath": the bit-exact checker (shares csrc/ptmath.h with the kernels); "libm": the independent build."""
        self.width, self.height = width, height
        self.n = width * height
        self._L = lib(LIB_LIBM if math == "libm" else LIB)
        self._c = self._L.oracle_create(C.byref(scene_desc), width, height, seed, max_iterations,
                                      1 if literal_slot_rng else 0, samples_per_pass)
        if not self._c:
            raise RuntimeError("oracle_create failed")
        self.pixels_host = np.zeros((self.n, 4), dtype=np.uint8)
        self.ticks = 1

    def close(self):
        if self._c:
            self._L.oracle_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def generate_frame(self, ticks=None):
        if ticks is None:
            ticks = self.ticks
            self.ticks += 1
        self._L.oracle_generate_frame(self._c, self.pixels_host.ctypes.data_as(C.c_void_p), ticks)

    def set_camera(self, cam):
        self._L.oracle_set_camera(self._c, C.byref(cam))

    def set_mode(self, use_path_tracer):
        self._L.oracle_set_mode(self._c, 1 if use_path_tracer else 0)

    def set_max_iterations(self, n):
        self._L.oracle_set_max_iterations(self._c, n)

    def request_reset(self):
        self._L.oracle_request_reset(self._c)

    def accumulator(self):
        return np.ctypeslib.as_array(self._L.oracle_accumulator(self._c), shape=(self.n, 3)).copy()

    def float_sum(self):
        return np.ctypeslib.as_array(self._L.oracle_float_sum(self._c), shape=(self.n, 3)).copy()

    def last_radiance0(self):
        return np.ctypeslib.as_array(self._L.oracle_last_radiance0(self._c), shape=(self.n, 3)).copy()

    def pixels(self):
        return self.pixels_host.copy()

    def live_counts(self):
        buf = (C.c_uint32 * 65)()
        n = self._L.oracle_live_counts(self._c, buf, 65)
        return np.array(buf[:n], dtype=np.uint32)

    def total_ray_bounces(self):
        return int(self._L.oracle_total_ray_bounces(self._c))

    def rng_state(self, pixel, lane=0):
        out = np.empty(6, dtype=np.uint32)
        self._L.oracle_rng_state(self._c, pixel, lane, out.ctypes.data_as(_u32p))
        return out

    def probe_shade(self, point, normal, material_idx, seed=1):
        out = (C.c_float * 3)()
        self._L.oracle_probe_shade(self._c, _f3(point), _f3(normal), material_idx, seed, out)
        return np.array(out[:], dtype=np.float32)


def probe_sphere(center, radius, origin, direction, max_distance=float("inf")):
    sp = Sphere()
    sp.position.x, sp.position.y, sp.position.z = center
    sp.radius = radius
    sp.materialIdx = 7
    ray = (C.c_float * 6)(*origin, *direction)
    out = (C.c_float * 8)()
    hit = lib().oracle_probe_sphere(C.byref(sp), ray, max_distance, out)
    return bool(hit), np.array(out[:], dtype=np.float32)


def probe_triangle(v0, v1, v2, origin, direction, max_distance=float("inf"), normals=None):
    t = Triangle()
    for name, v in (("vertex0", v0), ("vertex1", v1), ("vertex2", v2)):
        f = getattr(t, name)
        f.x, f.y, f.z = v
    normals = normals or [(0, 0, 1)] * 3
    for name, v in zip(("normal0", "normal1", "normal2"), normals):
        f = getattr(t, name)
        f.x, f.y, f.z = v
    t.materialIdx = 3
    ray = (C.c_float * 6)(*origin, *direction)
    out = (C.c_float * 8)()
    hit = lib().oracle_probe_triangle(C.byref(t), ray, max_distance, out)
    return bool(hit), np.array(out[:], dtype=np.float32)


def probe_fresnel(refr_index, cos_i):
    return float(lib().oracle_probe_fresnel(refr_index, cos_i))


def probe_rotate_y_to(target, v):
    out = (C.c_float * 3)()
    lib().oracle_probe_rotate_y_to(_f3(target), _f3(v), out)
    return np.array(out[:], dtype=np.float32)


def probe_sampler(kind, axis, param, seed, n):
    out = np.empty((n, 3), dtype=np.float32)
    lib().oracle_probe_sampler(kind, _f3(axis), param, seed, n, out.ctypes.data_as(_f32p))
    return out


def probe_quantize(radiance):
    return int(lib().oracle_probe_quantize(radiance))


def probe_rng(seed, sequence, n):
    state = np.empty(6, dtype=np.uint32)
    raw = np.empty(n, dtype=np.uint32)
    uni = np.empty(n, dtype=np.float32)
    lib().oracle_probe_rng(seed, sequence, n, state.ctypes.data_as(_u32p), raw.ctypes.data_as(_u32p),
                           uni.ctypes.data_as(_f32p))
    return state, raw, uni


def probe_eye_ray(x, y, width, height, cam, seed):
    out = (C.c_float * 6)()
    lib().oracle_probe_eye_ray(x, y, width, height, C.byref(cam), seed, out)
    return np.array(out[:], dtype=np.float32)
