"""tools/fuzz_parity.py [count=100] [first_seed=20000] — a longer run of the two scene fuzzers of tests/test_gpu_fuzz_scenes.py
(random records; axis-class triangles on a grid; 64-600 random spheres for the chunked traversal) against the oracle on one GPU: prints one line per mismatch and a total.
Test infrastructure (it calls the oracle); not part of the product."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "cuda-path-tracer-ss_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle  # noqa: E402
import ptss  # noqa: E402
from test_gpu_fuzz_scenes import axis_scene, random_scene  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0
for k in range(count):
    seed = first + k
    kind = ("random", "axis", "many-sphere")[k % 3]
    if kind == "many-sphere":   # 64..600 spheres: the chunked traversal (kd leaves, enclosing balls, ray-distance bound test, regrouped visits)
        scene, rng = random_scene(seed, ns=64 + (seed * 7919) % 537)
    else:
        scene, rng = (axis_scene if kind == "axis" else random_scene)(seed)
    w, h = int(rng.integers(32, 112)), int(rng.integers(24, 72))
    bounces, S = int(rng.integers(1, 10)), int(rng.choice([1, 2, 3]))
    one = int(rng.integers(0, 2))
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, samples_per_pass=S, seed=seed, one_launch_frames=one)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S, seed=seed)
    ok = True
    for _ in range(2):
        r.generate_frame()
        o.generate_frame()
        ok = ok and np.array_equal(r.live_counts(), o.live_counts())
    ok = ok and np.array_equal(r.accumulator(), o.accumulator()) and np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    ok = ok and r.total_ray_bounces() == o.total_ray_bounces() and r.guard_timeouts() == 0
    if not ok:
        bad += 1
        print("MISMATCH seed %d (%s) %dx%d bounces %d S %d one-launch %d" % (seed, kind, w, h, bounces, S, one), flush=True)
    if k % 20 == 19:
        print("%d scenes, %d mismatches" % (k + 1, bad), flush=True)
    r.close()
    o.close()
print("fuzz: %d scenes, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
