"""Scenes with 64 or more spheres take the chunked sphere traversal (spatially sorted 8-sphere chunks, conservative
chunk bounds, closest hit kept as (minimum distance, highest original index)). It must stay bit-identical to the
oracle, which tests every sphere in the caller's order: exact ties between duplicate spheres at distant indices, large
random scenes whose rays leave the unit-direction assumption (unnormalised vertex normals), the switch off, and a
camera far outside the range the structure is built for (the context then falls back to the plain image)."""
import os

import numpy as np
import pytest

import oracle
import ptss
from test_gpu_edge_scenes import CREAM, EMIT, GLASS, GREEN, MIRROR, RED, COOK, PHONG, FLOOR, LAMP, build
from test_gpu_fuzz_scenes import random_scene

pytestmark = pytest.mark.gpu


def run_pair(scene, w, h, bounces, ticks=2, S=1, seed=0x5EED, camera=None, every_sphere_loop=False):
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, samples_per_pass=S, seed=seed,
                      every_sphere_loop=every_sphere_loop)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S, seed=seed)
    if camera is not None:
        r.set_camera(camera)
        o.set_camera(camera)
    for _ in range(ticks):
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts())
    acc = r.accumulator()
    assert np.array_equal(acc, o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    r.close()
    return acc


def tie_scene():
    """96 spheres on a grid; every grid site holds the SAME sphere two or three times with different materials, the copies
    far apart in the array, so every primary hit is an exact tie that only the index order decides."""
    rng = np.random.default_rng(3)
    sites = [((-3.0 + 0.75 * (i % 9), -0.9 + 0.7 * (i // 9), -4.0 - 0.2 * (i % 3)), 0.3) for i in range(36)]
    mats = [RED, GREEN, CREAM, MIRROR, EMIT, COOK, PHONG, GLASS]
    spheres = []
    for rep in range(3):
        order = rng.permutation(len(sites)) if rep else np.arange(len(sites))
        for i in order[: 36 if rep < 2 else 24]:
            spheres.append((sites[i][0], sites[i][1], mats[int(rng.integers(0, len(mats)))]))
    assert len(spheres) == 96
    return build(spheres=spheres, triangles=FLOOR + LAMP, area=[((60, 60, 60), 2)], point=[((0, 2.5, -2), (30, 30, 30))])


def test_exact_ties_end_on_the_highest_index_like_the_sequential_loop():
    run_pair(tie_scene(), 96, 54, 5, ticks=3)


@pytest.mark.parametrize("seed,ns,nt", [(7001, 64, 6), (7002, 200, 20), (7003, 333, 0), (7004, 1000, 10)])
def test_large_random_scenes(seed, ns, nt):
    scene, rng = random_scene(seed, ns=ns, nt=nt)
    run_pair(scene, int(rng.integers(48, 100)), int(rng.integers(27, 60)), int(rng.integers(3, 9)), S=int(rng.choice([1, 3])), seed=seed)


@pytest.mark.parametrize("seed,ns", [(7101, 2300), (7102, 4500)])
def test_more_than_128_chunks(seed, ns):
    """Beyond 2,048 spheres the chunk bits come in batches of 128 chunks (four words per lane and batch), and the image no longer
    fits the LDS next to the work area: the kernels read it in place. Pair lists, camera-origin bound rows and the leaf
    refinement all see several batches here."""
    scene, rng = random_scene(seed, ns=ns, nt=8)
    run_pair(scene, 56, 32, 5, ticks=2, S=2, seed=seed)
    on = run_pair(scene, 40, 24, 4, ticks=1, seed=seed + 1)
    off = run_pair(scene, 40, 24, 4, ticks=1, seed=seed + 1, every_sphere_loop=True)
    assert np.array_equal(on, off)


def test_giant_wall_spheres_among_many_small_ones():
    """A smallpt-style room (walls = spheres of radius 1e5: chunk bounds a hundred thousand units wide, discriminants that
    cancel to b^2) filled with 90 small spheres: the chunked traversal and its shorter sphere test against the oracle."""
    R = 1e5
    rng = np.random.default_rng(11)
    mats = [RED, GREEN, CREAM, MIRROR, COOK, PHONG, GLASS]
    spheres = [((-R - 3, 0, -5), R, RED), ((R + 3, 0, -5), R, GREEN), ((0, 0, -R - 9), R, CREAM), ((0, -R - 2, -5), R, CREAM),
               ((0, R + 3, -5), R, CREAM)]
    for _ in range(90):
        spheres.append(((float(rng.uniform(-2.6, 2.6)), float(rng.uniform(-1.8, 2.6)), float(rng.uniform(-8.5, -2.5))),
                        float(rng.uniform(0.08, 0.3)), mats[int(rng.integers(0, len(mats)))]))
    scene = build(spheres=spheres, point=[((0, 2.7, -5), (70, 70, 70)), ((1.5, 0, -1.5), (20, 20, 20))])
    run_pair(scene, 72, 48, 7, ticks=2)
    run_pair(scene, 40, 30, 5, ticks=2, S=3)


def test_switch_off_gives_the_same_image():
    scene, _ = random_scene(7010, ns=300, nt=12)
    on = run_pair(scene, 64, 36, 6)
    off = run_pair(scene, 64, 36, 6, every_sphere_loop=True)   # cfg.everySphereLoop: the reference's loop over every sphere
    assert np.array_equal(on, off)


def test_stress_scene_at_720p_chunked_equals_unchunked():
    """BASELINE.json configs[5]'s scene, 12 bounces, at a size where millions of rays meet the chunk bounds: the chunked
    traversal against the same library with the structure switched off (that path is the one the oracle tests pin)."""
    def frames(every_sphere_loop=False):
        r = ptss.Renderer(ptss.Scene("stress"), 1280, 720, max_iterations=12, sync_each_frame=False, samples_per_pass=2,
                          every_sphere_loop=every_sphere_loop)
        counts = []
        for _ in range(3):
            r.generate_frame()
            counts.append(r.live_counts().copy())
        out = (r.accumulator(), np.array(counts), r.total_ray_bounces())
        r.close()
        return out
    on = frames()
    off = frames(every_sphere_loop=True)
    assert np.array_equal(on[1], off[1]) and on[2] == off[2]
    assert np.array_equal(on[0], off[0])


def test_config5_shape_4k_12_bounces_chunked_equals_unchunked():
    """The shape BASELINE.json configs[5] names — 3840x2160, 12 bounces, the 1,024-sphere scene — one pass of one sample
    per pixel: 8.3 million rays through the chunk bounds (and the regrouped traversal) against the every-sphere loop."""
    def frame(every_sphere_loop):
        r = ptss.Renderer(ptss.Scene("stress"), 3840, 2160, max_iterations=12, sync_each_frame=False,
                          every_sphere_loop=every_sphere_loop)
        r.generate_frame()
        out = (r.accumulator(), r.live_counts().copy(), r.total_ray_bounces())
        r.close()
        return out
    on, off = frame(False), frame(True)
    assert np.array_equal(on[1], off[1]) and on[2] == off[2]
    assert on[1][0] == 3840 * 2160 and len(on[1]) == 12
    assert np.array_equal(on[0], off[0])


def test_camera_outside_the_structures_range_falls_back():
    scene, _ = random_scene(7011, ns=128, nt=8)
    cam = ptss.default_camera()
    cam.position.x, cam.position.z = 3e16, 5.0          # beyond 1e15: squares still finite, but outside the proven range
    run_pair(scene, 40, 24, 4, camera=cam)
    cam.position.x = 0.25                                # and back inside: the chunked image again
    run_pair(scene, 40, 24, 4, camera=cam)


def test_stress_preset_switches_and_agrees():
    scene = ptss.Scene("stress")                         # BASELINE.json configs[5]: 1,024 spheres
    run_pair(scene, 80, 45, 6, ticks=2, S=2)


def test_camera_inside_the_cluster_rays_leaving_chunks_they_start_beside():
    """The chunk test is on the distance from the RAY (the half line), so a ray that starts beside or inside a chunk's bound
    and leaves it skips the chunk: with the camera in the middle of the 1,024 spheres every primary ray has most bounds
    behind it and many around it. Against the oracle at a small size, against the every-sphere loop at a larger one."""
    scene = ptss.Scene("stress")
    cam = ptss.default_camera()
    for pos in ((0.3, 0.2, -4.5), (-1.7, 1.1, -6.0), (2.4, -2.2, -2.6)):
        cam.position.x, cam.position.y, cam.position.z = pos
        run_pair(scene, 72, 40, 8, ticks=1, S=2, camera=cam)

    def frames(every_sphere_loop):
        r = ptss.Renderer(scene, 640, 360, max_iterations=12, sync_each_frame=False, samples_per_pass=2, every_sphere_loop=every_sphere_loop)
        r.set_camera(cam)
        for _ in range(2):
            r.generate_frame()
        out = (r.accumulator(), r.live_counts().copy(), r.total_ray_bounces())
        r.close()
        return out
    on, off = frames(False), frames(True)
    assert np.array_equal(on[1], off[1]) and on[2] == off[2]
    assert np.array_equal(on[0], off[0])
