"""tools/lanes_bench.py [config=c3] [S=1] [passes=200] [width height] — whole-frame Mrays/s of one context with 1, 2, 3, 4 frame
lanes, free-running (cfg.lanesFreeRun) and, for 2 lanes, ordered strictly on the caller's stream (width/height override the
config's frame: where does the automatic choice belong?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ptss  # noqa: E402
from bench import CONFIGS  # noqa: E402

cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
if len(sys.argv) > 5:
    cfg["width"], cfg["height"] = int(sys.argv[4]), int(sys.argv[5])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 200
scene = ptss.Scene(cfg["preset"])
for lanes, free in ((1, True), (2, True), (2, False), (3, True), (4, True), (1, True), (2, True), (2, False)):
    r = ptss.Renderer(scene, cfg["width"], cfg["height"], max_iterations=cfg["bounces"], sync_each_frame=False, samples_per_pass=S,
                      frame_lanes=lanes, lanes_free_run=free)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    pix = torch.zeros((r.local_pixels, 4), dtype=torch.uint8, device="cuda")
    for _ in range(max(3, passes // 10)):
        r.generate_frame(pix.data_ptr())
    torch.cuda.synchronize()
    r0 = r.total_ray_bounces()
    t0 = time.perf_counter()
    for _ in range(passes):
        r.generate_frame(pix.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = r.total_ray_bounces() - r0
    print("%s %dx%d S=%d lanes=%d %s: %.4f ms per pass, %.1f Mrays/s, guard timeouts %d" % (sys.argv[1] if len(sys.argv) > 1 else "c3", cfg["width"], cfg["height"], S, lanes, "free-running" if free else "strict",
          dt / passes * 1e3, rays / dt / 1e6, r.guard_timeouts()))
    r.close()
