"""BASELINE config 5's scene and size on ONE GPU (the config itself is an 8-GPU run): 3840x2160, preset 'stress'
(1,024 spheres + Cornell box), 12 bounces. Reports Mrays/s and per-bounce live counts for a few passes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
import ptss  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunked = (int(sys.argv[3]) != 0) if len(sys.argv) > 3 else True   # 0: cfg.everySphereLoop (the reference's loop)
r = ptss.Renderer(ptss.Scene("stress"), 3840, 2160, max_iterations=12, sync_each_frame=False, samples_per_pass=S,
                  every_sphere_loop=not chunked)
r.generate_frame()
r.synchronize()
r0 = r.total_ray_bounces()
t0 = time.perf_counter()
for _ in range(passes):
    r.generate_frame()
r.synchronize()
dt = time.perf_counter() - t0
rays = r.total_ray_bounces() - r0
print("stress 3840x2160, 1024 spheres, 12 bounces, S=%d: %.1f ms/pass, %.1f Mrays/s, live counts %s"
      % (S, dt / passes * 1e3, rays / dt / 1e6, r.live_counts().tolist()))
