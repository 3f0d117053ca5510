"""tools/cull_stat.py [bounces=12] — many-sphere scene (configs[5]'s): of the chunks whose bound a ray's line touches (what the
closest hit visits today), how many are NOT wholly beyond the ray's final hit distance (what a front-to-back order would
have to visit at most)? Diagnostic build: python tools/build_variants.py cullstat; PTSS_LIBNAME=libptss_cullstat.so python tools/cull_stat.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

bounces = int(sys.argv[1]) if len(sys.argv) > 1 else 12
r = ptss.Renderer(ptss.Scene("stress"), 1920, 1080, max_iterations=bounces, sync_each_frame=False, samples_per_pass=1)
r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_counters(r._ctx, out) == 0
touched, needed, rays, hits = out[0], out[1], out[2], out[3]
print("%d bounces: rays %d, sphere hits %.1f %%, chunks touched per ray %.2f, not beyond the final hit %.2f (%.0f %%)"
      % (bounces, rays, 100.0 * hits / rays, touched / rays, needed / rays, 100.0 * needed / max(touched, 1)))
r.close()
