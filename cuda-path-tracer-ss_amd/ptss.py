"""ptss.py — thin ctypes binding of the two product libraries for the Python harness
(tests/, bench.py, __graft_entry__.py).

  libptss_host.so  include/ptss_host.h  host mirror (Scene presets, camera, TGA, tile rows, probes)
  libptss.so       include/ptss.h       the HIP hot path (no CPU fallback: create() raises without a GPU)

PyTorch is used by callers only for device memory / streams / torch.distributed; nothing here
imports torch.
"""
import ctypes as C
import os

import numpy as np

from ptss_types import (AreaLight, Camera, Material, PointLight, SceneDesc, Sphere, Triangle, UChar4, Vec3,
                        struct_to_dict)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(_HERE, "lib")
HOST_LIB = os.path.join(LIBDIR, "libptss_host.so")
DEVICE_LIB = os.path.join(LIBDIR, os.environ.get("PTSS_LIBNAME", "libptss.so"))  # PTSS_LIBNAME: A/B variants

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


class PtssError(RuntimeError):
    pass


class RenderConfig(C.Structure):
    _fields_ = [("structSize", C.c_uint), ("width", C.c_int), ("height", C.c_int), ("seed", C.c_ulonglong), ("maxIterations", C.c_uint),
                ("device", C.c_int), ("tileRank", C.c_int), ("tileWorld", C.c_int), ("bandRows", C.c_int),
                ("syncEachFrame", C.c_int), ("floatAccumulator", C.c_int), ("timeKernels", C.c_int),
                ("samplesPerPass", C.c_int), ("everySphereLoop", C.c_int), ("frameLanes", C.c_int),
                ("lanesFreeRun", C.c_int), ("oneLaunchFrames", C.c_int)]


_host = None
_dev = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise PtssError(f"{HOST_LIB} missing: run `python __graft_entry__.py build` first")
        L = C.CDLL(HOST_LIB)
        L.ptss_scene_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptss_scene_destroy.argtypes = [C.c_void_p]
        L.ptss_scene_destroy.restype = None
        L.ptss_scene_describe.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.ptss_camera_default.argtypes = [C.POINTER(Camera)]
        L.ptss_camera_move.argtypes = [C.POINTER(Camera), C.c_ubyte, C.POINTER(C.c_int)]
        L.ptss_write_tga.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.ptss_tile_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.ptss_probe_math.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_size_t]
        L.ptss_probe_quantize.argtypes = [_f32p, C.POINTER(C.c_uint), C.c_size_t]
        L.ptss_probe_triangle_forms.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_size_t, C.POINTER(C.c_int), _f32p, _f32p]
        L.ptss_probe_quant_table.argtypes = [_f32p]
        L.ptss_probe_rng_init.argtypes = [C.c_ulonglong, C.c_uint, _u32p]
        L.ptss_probe_rng_draw.argtypes = [_u32p, _u32p, _f32p, C.c_size_t]
        L.ptss_probe_rng_jump_table.argtypes = [_u32p, C.c_size_t]
        _host = L
    return _host


def _torch_first():
    """PyTorch bundles its own HIP runtime and libptss.so links ROCm's; both may share a process, but torch's must initialise
    first or a later torch.cuda call can fail with "No HIP GPUs are available" (INTEGRATION.md §5). If torch is already
    imported, initialise it now — before libptss.so's runtime touches the device."""
    import sys
    torch = sys.modules.get("torch")
    if torch is None:
        return
    try:
        if torch.cuda.is_available() and not torch.cuda.is_initialized():
            torch.cuda.init()
    except Exception as e:  # pragma: no cover - depends on the box
        raise PtssError("torch is imported but torch.cuda could not be initialised before libptss.so's HIP runtime "
                        f"(INTEGRATION.md §5, two HIP runtimes in one process): {e}")


def device_lib():
    """Loads libptss.so. Loading needs no GPU; ptss_create does."""
    global _dev
    _torch_first()
    if _dev is None:
        if not os.path.exists(DEVICE_LIB):
            raise PtssError(f"{DEVICE_LIB} missing: the HIP extension was not built (no CPU fallback exists)")
        L = C.CDLL(DEVICE_LIB)
        vp = C.c_void_p
        L.ptss_default_config.argtypes = [C.POINTER(RenderConfig)]
        L.ptss_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(RenderConfig), C.POINTER(vp)]
        L.ptss_destroy.argtypes = [vp]
        L.ptss_generate_frame.argtypes = [vp, vp, C.c_int]
        L.ptss_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.ptss_get_camera.argtypes = [vp, C.POINTER(Camera)]
        L.ptss_request_reset.argtypes = [vp]
        L.ptss_set_mode.argtypes = [vp, C.c_int]
        L.ptss_set_max_iterations.argtypes = [vp, C.c_uint]
        L.ptss_set_stream.argtypes = [vp, vp]
        L.ptss_bind_accumulator.argtypes = [vp, vp]
        L.ptss_accumulator_devptr.argtypes = [vp, C.POINTER(vp)]
        L.ptss_float_accumulator_devptr.argtypes = [vp, C.POINTER(vp)]
        L.ptss_alloc_pixels.argtypes = [vp, C.POINTER(vp)]
        L.ptss_free_pixels.argtypes = [vp, vp]
        L.ptss_local_pixels.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.ptss_local_rows.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.ptss_read_accumulator.argtypes = [vp, _u32p, C.c_size_t]
        L.ptss_read_float_accumulator.argtypes = [vp, _f32p, C.c_size_t]
        L.ptss_read_pixels.argtypes = [vp, vp, vp, C.c_size_t]
        L.ptss_read_rng_state.argtypes = [vp, C.c_size_t, _u32p]
        L.ptss_read_rng_state_lane.argtypes = [vp, C.c_size_t, C.c_uint, _u32p]
        L.ptss_synchronize.argtypes = [vp]
        L.ptss_last_pass_ms.argtypes = [vp, _f32p]
        L.ptss_samples_since_reset.argtypes = [vp, C.POINTER(C.c_int)]
        L.ptss_live_counts.argtypes = [vp, _u32p, C.c_int, C.POINTER(C.c_int)]
        L.ptss_total_ray_bounces.argtypes = [vp, C.POINTER(C.c_ulonglong)]
        L.ptss_frame_lanes.argtypes = [vp, C.POINTER(C.c_int)]
        L.ptss_one_launch_frames.argtypes = [vp, C.POINTER(C.c_int)]
        L.ptss_guard_timeouts.argtypes = [vp, C.POINTER(C.c_uint)]
        L.ptss_bounce_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]
        L.ptss_error_string.argtypes = [C.c_int]
        L.ptss_error_string.restype = C.c_char_p
        L.ptss_last_error_detail.restype = C.c_char_p
        _dev = L
    return _dev


def _check(rc):
    if rc != 0:
        L = device_lib()
        raise PtssError(f"ptss error {rc} ({L.ptss_error_string(rc).decode()}): {L.ptss_last_error_detail().decode()}")


class Scene:
    """class Scene of the host mirror (reference: CudaTracer/Scene.h:5-27), built from a preset name."""

    def __init__(self, preset="default"):
        self._h = C.c_void_p()
        rc = host_lib().ptss_scene_create(preset.encode(), C.byref(self._h))
        if rc != 0:
            raise PtssError(f"unknown scene preset {preset!r}")
        self.preset = preset
        self.desc = SceneDesc()
        host_lib().ptss_scene_describe(self._h, C.byref(self.desc))
        self.desc._owner = self  # desc borrows the scene's arrays: `Scene(p).desc` must keep the scene alive

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and host_lib is not None:
            host_lib().ptss_scene_destroy(self._h)
            self._h = C.c_void_p()

    def _arr(self, ptr, n):
        return [ptr[i] for i in range(n)]

    @property
    def spheres(self):
        return self._arr(self.desc.spheres, self.desc.numSpheres)

    @property
    def triangles(self):
        return self._arr(self.desc.triangles, self.desc.numTriangles)

    @property
    def materials(self):
        return self._arr(self.desc.materials, self.desc.numMaterials)

    @property
    def area_lights(self):
        return self._arr(self.desc.areaLights, self.desc.numAreaLights)

    @property
    def point_lights(self):
        return self._arr(self.desc.pointLights, self.desc.numPointLights)

    def table(self):
        """Plain-Python dump of every record (the committed scene fixtures)."""
        return {
            "spheres": [struct_to_dict(s) for s in self.spheres],
            "triangles": [struct_to_dict(s) for s in self.triangles],
            "materials": [struct_to_dict(s) for s in self.materials],
            "pointLights": [struct_to_dict(s) for s in self.point_lights],
            "areaLights": [struct_to_dict(s) for s in self.area_lights],
        }


def default_camera():
    cam = Camera()
    host_lib().ptss_camera_default(C.byref(cam))
    return cam


def move_camera(cam, key):
    moved = C.c_int(0)
    host_lib().ptss_camera_move(C.byref(cam), ord(key), C.byref(moved))
    return bool(moved.value)


def tile_rows(height, band_rows, rank, world):
    n = host_lib().ptss_tile_rows(height, band_rows, rank, world, None, 0)
    if n < 0:
        raise PtssError("bad tile spec")
    rows = (C.c_int * max(n, 1))()
    host_lib().ptss_tile_rows(height, band_rows, rank, world, rows, n)
    return np.array(rows[:n], dtype=np.int64)


def write_tga(path, rgba_hw4):
    a = np.ascontiguousarray(rgba_hw4, dtype=np.uint8)
    h, w = a.shape[:2]
    rc = host_lib().ptss_write_tga(path.encode(), a.ctypes.data_as(C.c_void_p), w, h)
    if rc != 0:
        raise PtssError(f"write_tga failed ({rc})")


class Renderer:
    """One ptss_context: the reference's ProgramData + device buffers, driven like generateFrame."""

    def __init__(self, scene, width, height, max_iterations=15, seed=0x5EED, device=0, tile_rank=0, tile_world=1,
                 band_rows=8, sync_each_frame=True, float_accumulator=False, time_kernels=False, samples_per_pass=1,
                 every_sphere_loop=False, frame_lanes=0, lanes_free_run=False, one_launch_frames=0):
        L = device_lib()
        cfg = RenderConfig()
        _check(L.ptss_default_config(C.byref(cfg)))
        cfg.width, cfg.height = width, height
        cfg.seed = seed
        cfg.maxIterations = max_iterations
        cfg.device = device
        cfg.tileRank, cfg.tileWorld, cfg.bandRows = tile_rank, tile_world, band_rows
        cfg.syncEachFrame = 1 if sync_each_frame else 0
        cfg.floatAccumulator = 1 if float_accumulator else 0
        cfg.timeKernels = 1 if time_kernels else 0
        cfg.samplesPerPass = samples_per_pass
        cfg.everySphereLoop = 1 if every_sphere_loop else 0
        cfg.frameLanes = frame_lanes
        cfg.lanesFreeRun = 1 if lanes_free_run else 0
        cfg.oneLaunchFrames = one_launch_frames
        self.cfg = cfg
        self._scene = scene  # keep the arrays alive during create
        self._ctx = C.c_void_p()
        _check(L.ptss_create(C.byref(scene.desc), C.byref(cfg), C.byref(self._ctx)))
        n = C.c_size_t()
        _check(L.ptss_local_pixels(self._ctx, C.byref(n)))
        self.local_pixels = n.value
        self.width, self.height = width, height
        self.local_rows = self.local_pixels // width
        self._own_pixels = None
        self.ticks = 1  # GPUAnimBitmap::idle_func's static counter starts at 1 (CudaUtils.h:146)

    def close(self):
        if self._ctx:
            L = device_lib()
            if self._own_pixels:
                L.ptss_free_pixels(self._ctx, self._own_pixels)
                self._own_pixels = None
            L.ptss_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- plumbing -------------------------------------------------------------------------------
    def pixels_devptr(self):
        if self._own_pixels is None:
            p = C.c_void_p()
            _check(device_lib().ptss_alloc_pixels(self._ctx, C.byref(p)))
            self._own_pixels = p
        return self._own_pixels

    def set_stream(self, raw_stream):
        _check(device_lib().ptss_set_stream(self._ctx, C.c_void_p(raw_stream)))

    def bind_accumulator(self, devptr):
        _check(device_lib().ptss_bind_accumulator(self._ctx, C.c_void_p(devptr)))

    def rows(self):
        cnt = C.c_int()
        rows = (C.c_int * max(self.local_rows, 1))()
        _check(device_lib().ptss_local_rows(self._ctx, rows, self.local_rows, C.byref(cnt)))
        return np.array(rows[:cnt.value], dtype=np.int64)

    # --- the frame callback ---------------------------------------------------------------------
    def generate_frame(self, dev_pixels=None, ticks=None):
        """generateFrame(pixels, dataBlock, ticks). With no arguments behaves like one GLUT idle tick."""
        if ticks is None:
            ticks = self.ticks
            self.ticks += 1
        if dev_pixels is None:
            dev_pixels = self.pixels_devptr()
        elif isinstance(dev_pixels, int):
            dev_pixels = C.c_void_p(dev_pixels)
        _check(device_lib().ptss_generate_frame(self._ctx, dev_pixels, ticks))

    def set_camera(self, cam):
        _check(device_lib().ptss_set_camera(self._ctx, C.byref(cam)))

    def get_camera(self):
        cam = Camera()
        _check(device_lib().ptss_get_camera(self._ctx, C.byref(cam)))
        return cam

    def request_reset(self):
        _check(device_lib().ptss_request_reset(self._ctx))

    def set_mode(self, use_path_tracer):
        _check(device_lib().ptss_set_mode(self._ctx, 1 if use_path_tracer else 0))

    def set_max_iterations(self, n):
        _check(device_lib().ptss_set_max_iterations(self._ctx, n))

    def synchronize(self):
        _check(device_lib().ptss_synchronize(self._ctx))

    # --- read-back ------------------------------------------------------------------------------
    def accumulator(self):
        out = np.empty((self.local_pixels, 3), dtype=np.uint32)
        _check(device_lib().ptss_read_accumulator(self._ctx, out.ctypes.data_as(_u32p), out.size))
        return out

    def float_accumulator(self):
        out = np.empty((self.local_pixels, 3), dtype=np.float32)
        _check(device_lib().ptss_read_float_accumulator(self._ctx, out.ctypes.data_as(_f32p), out.size))
        return out

    def pixels(self, dev_pixels=None):
        if dev_pixels is None:
            dev_pixels = self.pixels_devptr()
        elif isinstance(dev_pixels, int):
            dev_pixels = C.c_void_p(dev_pixels)
        out = np.empty((self.local_pixels, 4), dtype=np.uint8)
        _check(device_lib().ptss_read_pixels(self._ctx, dev_pixels, out.ctypes.data_as(C.c_void_p), self.local_pixels))
        return out

    def rng_state(self, local_pixel, lane=0):
        out = np.empty(6, dtype=np.uint32)
        _check(device_lib().ptss_read_rng_state_lane(self._ctx, local_pixel, lane, out.ctypes.data_as(_u32p)))
        return out

    def last_pass_ms(self):
        v = C.c_float()
        _check(device_lib().ptss_last_pass_ms(self._ctx, C.byref(v)))
        return v.value

    def live_counts(self):
        out = (C.c_uint32 * 65)()
        n = C.c_int()
        _check(device_lib().ptss_live_counts(self._ctx, out, 65, C.byref(n)))
        return np.array(out[:n.value], dtype=np.uint32)

    def total_ray_bounces(self):
        v = C.c_ulonglong()
        _check(device_lib().ptss_total_ray_bounces(self._ctx, C.byref(v)))
        return v.value

    @property
    def one_launch_frames(self):
        """True when this context traces a frame with ONE launch (cfg.oneLaunchFrames resolved for the current scene image)."""
        v = C.c_int()
        _check(device_lib().ptss_one_launch_frames(self._ctx, C.byref(v)))
        return bool(v.value)

    @property
    def frame_lanes(self):
        v = C.c_int()
        _check(device_lib().ptss_frame_lanes(self._ctx, C.byref(v)))
        return v.value

    def guard_timeouts(self):
        v = C.c_uint()
        _check(device_lib().ptss_guard_timeouts(self._ctx, C.byref(v)))
        return v.value

    def bounce_kernel_time(self):
        ms = C.c_double()
        n = C.c_ulonglong()
        _check(device_lib().ptss_bounce_kernel_time(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value
