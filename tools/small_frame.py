"""tools/small_frame.py [width=512] [height=512] [passes=400] [preset=cornell] [bounces=15] [S=1] [one_launch=0|1] — the reference's own mode at its own size
(DIM = 512, one sample per tick, maxIterations 15): ms per pass and Mrays/s of one context; run under
rocprofv3 --kernel-trace + tools/trace_gaps.py to see kernel durations against the gaps between them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
preset = sys.argv[4] if len(sys.argv) > 4 else "cornell"
bounces = int(sys.argv[5]) if len(sys.argv) > 5 else 15
S = int(sys.argv[6]) if len(sys.argv) > 6 else 1
one = int(sys.argv[7]) if len(sys.argv) > 7 else 0   # cfg.oneLaunchFrames: 1 = one launch per frame where the frame qualifies, 0 = bounce by bounce
r = ptss.Renderer(ptss.Scene(preset), w, h, max_iterations=bounces, sync_each_frame=False, samples_per_pass=S, one_launch_frames=one)
for _ in range(50):
    r.generate_frame()
r.synchronize()
r0 = r.total_ray_bounces()
t0 = time.perf_counter()
for _ in range(passes):
    r.generate_frame()
r.synchronize()
dt = time.perf_counter() - t0
print("%s %dx%d %d bounces S=%d lanes=%d one-launch=%s: %.4f ms per pass, %.1f Mrays/s, live %s" % (preset, w, h, bounces, S, r.frame_lanes, r.one_launch_frames, dt / passes * 1e3,
      (r.total_ray_bounces() - r0) / dt / 1e6, r.live_counts()))
r.close()
