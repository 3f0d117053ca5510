"""Which blocks of scatter() do the waves execute, and for how many of their lanes? (diagnostic build libptss_shist.so)
   PTSS_LIBNAME=libptss_shist.so python tools/scatter_hist.py [S]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = ptss.Renderer(ptss.Scene("mixed"), 1920, 1080, max_iterations=8, sync_each_frame=False, samples_per_pass=S, frame_lanes=1)
for _ in range(4):
    r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_counters(r._ctx, out) == 0
names = ["scatter at all", "non-Lambert block (Snell, specular, refraction)", "Snell / Fresnel terms", "refraction lobe", "sampler tail (2 draws, sincos, rotation)",
         "... Beckmann elevation + Cook-Torrance weight", "... Phong elevation (pow)", "... Lambert elevation (sqrt)"]
waves0 = out[0] & 0xffffffff
for n, v in zip(names, out):
    waves, lanes = v & 0xffffffff, v >> 32
    print("%-52s %6.1f %% of the waves in scatter, %5.1f lanes of 64 each time" % (n, 100.0 * waves / max(waves0, 1), lanes / max(waves, 1)))
