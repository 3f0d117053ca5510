"""Sphere candidates per lane in the closest-hit and any-hit loops: wave maximum (what the wave pays) against the wave
mean (what a perfectly regrouped evaluation would pay). Diagnostic build: PTSS_LIBNAME=libptss_chist.so python tools/candidate_hist.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

r = ptss.Renderer(ptss.Scene("mixed"), 1920, 1080, max_iterations=8, sync_each_frame=False, samples_per_pass=4)
for _ in range(4):
    r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_counters(r._ctx, out) == 0
for name, o in (("closest hit", 0), ("any hit", 4)):
    mx, total, n, live = out[o], out[o + 1], out[o + 2], out[o + 3]
    print("%-12s chunks %10d  live lanes/wave %5.1f  candidates: wave max %.2f, mean per lane %.2f (per live lane %.2f) -> %.0f %% of the candidate loop is idle lanes"
          % (name, n, live / n, mx / n, total / n / 64, total / max(live, 1), 100 * (1 - total / 64 / max(mx, 1))))
