"""Histogram of the shadow-ray queue length per wave per NEE round (diagnostic build libptss_qhist.so):
   PTSS_LIBNAME=libptss_qhist.so python tools/queue_hist.py [S]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = ptss.Renderer(ptss.Scene("mixed"), 1920, 1080, max_iterations=8, sync_each_frame=False, samples_per_pass=S)
for _ in range(6):
    r.generate_frame()
r.synchronize()
L = ptss.device_lib()
L.ptss_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 8)()
assert L.ptss_debug_counters(r._ctx, out) == 0
names = ["0", "1-8", "9-16", "17-32", "33-64", "65-72", "73-96", "97-128"]
passes_now = [0, 1, 1, 1, 1, 2, 2, 2]
# split passes: q <= 32 gets g = 64 / pow2ceil(q) lanes per entry; a second pass holds q - 64 entries
passes_split = [0, 1 / 8, 1 / 4, 1 / 2, 1, 1 + 1 / 8, 1 + 1 / 2, 2]
tot = sum(out)
for n, v in zip(names, out):
    print("queue %-7s %6.2f %% of rounds" % (n, 100.0 * v / max(tot, 1)))
now = sum(p * v for p, v in zip(passes_now, out))
new = sum(p * v for p, v in zip(passes_split, out))
print("dense passes now: %.3f per round; with lane-split sparse passes: ~%.3f (%.1f %% less)" % (now / tot, new / tot, 100 * (1 - new / now)))
