"""Per-phase wave-cycle shares of the bounce kernel (diagnostic build libptss_stamps.so).
   PTSS_LIBNAME=libptss_stamps.so python tools/phase_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

names = ["ray load", "closest hit", "surfel + light sampling + enqueue", "dense shadow passes",
         "lambert + scatter + update", "finish (tonemap/accumulate/park)", "compaction", "-"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # sample lanes per pixel per pass
r = ptss.Renderer(ptss.Scene("mixed"), 1920, 1080, max_iterations=8, sync_each_frame=False, samples_per_pass=S)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_phase_cycles(r._ctx, out) == 0
tot = sum(out)
rays = r.total_ray_bounces()
print("total wave-cycles", tot, " per 64 rays:", tot / (rays / 64.0))
for n, v in zip(names, out):
    print("%-36s %6.2f %%   %9.0f cycles per wave-tile" % (n, 100.0 * v / max(tot, 1), v / (rays / 64.0)))
