"""One launch per frame (cfg.oneLaunchFrames, ptss_kernels.hip frameKernel): a frame whose bounce-0 tiles are all resident at
once is traced by ONE kernel whose workgroups carry their shard from bounce to bounce (per-shard done counters, rays handed
over through sc1 accesses) instead of one launch per bounce. Image, counters and random streams must be the oracle's — and
the bounce-by-bounce path's — bit for bit, including the whole-frame loop guard `numRays > 128` (CudaTracer.cu:622), which a
workgroup whose shard holds <= 128 rays decides by following the other shards' counters."""
import numpy as np
import pytest

import oracle
import ptss

pytestmark = pytest.mark.gpu


def run_pair(preset, w, h, bounces, ticks, S=1, seed=0x5EED, one_launch=1, expect=True):
    scene = ptss.Scene(preset)
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, seed=seed, float_accumulator=True, samples_per_pass=S, one_launch_frames=one_launch)
    assert r.one_launch_frames == expect
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, seed=seed, samples_per_pass=S)
    for t in range(ticks):
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts()), t
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    for p in (0, w * h // 2, w * h - 1):
        assert np.array_equal(r.rng_state(p), o.rng_state(p))
    assert r.guard_timeouts() == 0
    r.close()
    o.close()


@pytest.mark.parametrize("preset,w,h,bounces,S", [("cornell", 64, 64, 5, 1), ("mixed", 100, 37, 8, 1), ("mixed", 48, 27, 6, 3), ("lambert", 120, 68, 8, 1),
                                                  ("default", 96, 96, 15, 1), ("pointlight", 80, 45, 6, 2), ("mixed", 256, 144, 8, 1)])
def test_one_launch_frames_match_the_oracle(preset, w, h, bounces, S):
    run_pair(preset, w, h, bounces, 3, S=S)
    run_pair(preset, w, h, bounces, 2, S=S, one_launch=0, expect=False)   # ... and so does the bounce-by-bounce path at the same size


@pytest.mark.parametrize("w,h,bounces", [(16, 8, 4), (8, 8, 3), (20, 10, 12), (24, 16, 15), (40, 20, 15), (64, 48, 15), (33, 31, 9)])
def test_loop_guard_is_exact_in_one_launch(w, h, bounces):
    """Frames so small that the frame-wide live count falls to <= 128 at bounce 0 (128 and 64 pixels: nothing runs) or somewhere
    along the path: every workgroup must stop exactly where the reference's host loop stops (the oracle's live counts show
    where), which a workgroup of a nearly empty shard learns from the other shards' counters."""
    run_pair("cornell", w, h, bounces, 4)
    run_pair("mixed", w, h, bounces, 2, S=2)


def test_reference_size_one_sample_per_tick():
    """The reference's own configuration: 512 x 512, 15 bounces, one sample per tick (CudaUtils.h:7, CudaTracer.h:39) — 1,024
    workgroups resident at once, fifteen bounces in one launch; against the bounce-by-bounce path over 40 ticks."""
    scene = ptss.Scene("default")
    out = {}
    for mode in (1, 0):
        r = ptss.Renderer(scene, 512, 512, max_iterations=15, sync_each_frame=False, one_launch_frames=mode)
        assert r.one_launch_frames == (mode == 1)
        for _ in range(40):
            r.generate_frame()
        out[mode] = (r.accumulator(), r.pixels(), r.live_counts().copy(), r.total_ray_bounces())
        assert r.guard_timeouts() == 0
        r.close()
    assert np.array_equal(out[1][2], out[0][2]) and out[1][3] == out[0][3]
    assert np.array_equal(out[1][0], out[0][0])
    assert np.array_equal(out[1][1], out[0][1])
    o = oracle.Oracle(scene.desc, 512, 512, max_iterations=15)
    for _ in range(40):
        o.generate_frame()
    assert np.array_equal(out[1][0], o.accumulator())


def test_mode_and_bounce_count_changes_between_one_launch_frames():
    """Ray-tracing mode (one bounce: bounce 0 is also the last), back, other bounce counts: the done counters are re-armed by
    flushKernel with the counts, frame after frame."""
    scene = ptss.Scene("mixed")
    w, h = 72, 40
    r = ptss.Renderer(scene, w, h, max_iterations=8, one_launch_frames=1)
    assert r.one_launch_frames
    o = oracle.Oracle(scene.desc, w, h, max_iterations=8)
    cam = ptss.default_camera()
    for step in range(11):
        if step == 2:
            r.set_mode(False); o.set_mode(False)
        if step == 4:
            r.set_mode(True); o.set_mode(True)
        if step == 6:
            r.set_max_iterations(3); o.set_max_iterations(3)
        if step == 7:
            r.set_max_iterations(11); o.set_max_iterations(11)
        if step == 9:
            cam.position.x = 0.25
            r.set_camera(cam); o.set_camera(cam)
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts()), step
        assert np.array_equal(r.accumulator(), o.accumulator()), step
    assert np.array_equal(r.pixels(), o.pixels())
    assert r.guard_timeouts() == 0
    r.close()


def test_many_spheres_in_one_launch():
    """The chunked many-sphere image (its own frame-kernel instantiation, lower residency) at a size that still qualifies."""
    scene = ptss.Scene("stress")
    w, h, bounces = 160, 90, 12
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, one_launch_frames=1)
    assert r.one_launch_frames
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces)
    for _ in range(2):
        r.generate_frame()
        o.generate_frame()
    assert np.array_equal(r.live_counts(), o.live_counts())
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.guard_timeouts() == 0
    r.close()
    o.close()


def test_which_frames_qualify():
    scene = ptss.Scene("cornell")
    off = ptss.Renderer(scene, 512, 512)      # opt-in: the default configuration traces bounce by bounce
    assert not off.one_launch_frames
    off.close()
    for w, h, S, lanes, want in ((512, 512, 1, 0, True), (640, 480, 1, 0, True), (1920, 1080, 1, 0, False),   # 8,100 tiles: not resident at once
                                 (512, 512, 8, 0, False),                                                    # 8,192 tiles
                                 (256, 256, 4, 0, True), (512, 512, 1, 2, False)):                            # two frame lanes: bounce by bounce
        r = ptss.Renderer(scene, w, h, samples_per_pass=S, frame_lanes=lanes, one_launch_frames=1)
        assert r.one_launch_frames == want, (w, h, S, lanes)
        r.close()
