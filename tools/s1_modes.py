"""tools/s1_modes.py <mode: lane1|lanes2|shards2> [passes=300] [config=c3] — the reference's mode (one sample per tick) on the config's
frame in one of three ways: one context with one lane, one context with two free-running frame lanes (cfg.lanesFreeRun), two pixel-band shard contexts on two
streams. One mode per process, so that a rocprofv3 --kernel-trace of it (tools/lanes_trace.py) shows that mode alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-ss_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ptss  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "c3"]
scene = ptss.Scene(cfg["preset"])
shards, lanes = {"lane1": (1, 1), "lanes2": (1, 2), "shards2": (2, 1)}[mode]
v, ms = bench.s1_leg(ptss, torch, scene, cfg, shards=shards, passes=passes, frame_lanes=lanes, lanes_free_run=True)
print("%s: %.1f Mrays/s, %.4f ms per pass" % (mode, v, ms))
