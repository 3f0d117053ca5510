"""tools/lanes_large.py — one against two free-running frame lanes (cfg.lanesFreeRun) on the two wide configurations (c5's 4K frame at
S = 4, c3's 1080p frame at S = 40), each measured twice, interleaved, in one process on one box; the last column is what the library
picks by itself (frame_lanes = 0). Behind the upper limit of the automatic two-lane rule in ptss_api.hip."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

for preset, w, h, b, S in (("stress", 3840, 2160, 12, 4), ("mixed", 1920, 1080, 8, 40)):
    for rep in (1, 2):
        for lanes in (1, 2, 0):
            r = ptss.Renderer(ptss.Scene(preset), w, h, max_iterations=b, sync_each_frame=False, samples_per_pass=S, frame_lanes=lanes, lanes_free_run=True)
            for _ in range(4):
                r.generate_frame()
            r.synchronize()
            r0 = r.total_ray_bounces()
            t = time.perf_counter()
            n = 16
            for _ in range(n):
                r.generate_frame()
            r.synchronize()
            dt = time.perf_counter() - t
            print("%s S=%d rep %d lanes asked %d, run %d: %.1f Mrays/s %.2f ms/pass, guard timeouts %d" % (
                preset, S, rep, lanes, r.frame_lanes, (r.total_ray_bounces() - r0) / dt / 1e6, dt / n * 1e3, r.guard_timeouts()), flush=True)
            r.close()
