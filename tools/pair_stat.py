"""tools/pair_stat.py [preset=mixed] — per NEE round and wave: how many lit lanes need BOTH of the round's shadow segments, how many
exactly one (what decides whether the paired any-hit pays). Diagnostic build: python tools/build_variants.py pairstat;
PTSS_LIBNAME=libptss_pairstat.so python tools/pair_stat.py [preset]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "mixed"
r = ptss.Renderer(ptss.Scene(preset), 1920, 1080, max_iterations=8, sync_each_frame=False, samples_per_pass=2)
for _ in range(2):
    r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_counters(r._ctx, out) == 0
both, one, lit, rounds = out[4], out[5], out[6], out[7]
print("%s: per wave and round: lit lanes %.1f, both segments %.1f, one segment %.1f (%.0f %% of the lanes that queue anything)"
      % (preset, lit / rounds, both / rounds, one / rounds, 100.0 * one / max(both + one, 1)))
r.close()
