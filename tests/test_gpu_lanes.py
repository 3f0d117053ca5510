"""Frame lanes (cfg.frameLanes; ptss_device.h "frame lanes"): one frame traced as K ray populations on K streams of the
device. The image must not depend on K — including the reference's whole-frame loop guard `numRays > 128`
(CudaTracer.cu:622), which a lane holding <= 128 rays decides by asking its peer lanes for their counts."""
import numpy as np
import pytest

import oracle
import ptss

pytestmark = pytest.mark.gpu


def run_pair(preset, w, h, bounces, ticks, lanes, S=1, seed=0x5EED, free_run=False):
    scene = ptss.Scene(preset)
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, seed=seed, float_accumulator=True, samples_per_pass=S, frame_lanes=lanes,
                      lanes_free_run=free_run)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, seed=seed, samples_per_pass=S)
    assert r.frame_lanes == lanes
    for _ in range(ticks):
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts())
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    for p in (0, w * h // 2, w * h - 1):
        assert np.array_equal(r.rng_state(p), o.rng_state(p))
    assert r.guard_timeouts() == 0
    r.close()
    o.close()


@pytest.mark.parametrize("lanes", [2, 3, 4])
@pytest.mark.parametrize("preset,w,h,bounces,S", [("cornell", 64, 64, 5, 1), ("mixed", 100, 37, 8, 1), ("mixed", 48, 27, 6, 3)])
def test_lanes_match_the_oracle(lanes, preset, w, h, bounces, S):
    run_pair(preset, w, h, bounces, 3, lanes, S=S)
    run_pair(preset, w, h, bounces, 3, lanes, S=S, free_run=True)


@pytest.mark.parametrize("lanes", [2, 4])
@pytest.mark.parametrize("w,h,bounces", [(16, 8, 4), (8, 8, 3), (20, 10, 12), (24, 16, 15), (40, 20, 15)])
def test_loop_guard_is_exact_across_lanes(lanes, w, h, bounces):
    """Frames so small that the frame-wide live count falls to <= 128 at bounce 0 (128 and 64 pixels: nothing runs) or
    somewhere along the path: every lane must stop exactly where the single population stops (the oracle's live counts
    show where), so lanes that hold <= 128 rays have to add up their peers' counters."""
    run_pair("cornell", w, h, bounces, 4, lanes, free_run=True)
    run_pair("mixed", w, h, bounces, 2, lanes, S=2)


def test_mode_and_bounce_count_changes_with_lanes():
    """The two count buffers of a lane alternate per frame; a change of the bounce count between frames must not leave
    stale counts behind (ray-tracing mode = 1 bounce, then back)."""
    scene = ptss.Scene("mixed")
    w, h = 72, 40
    r = ptss.Renderer(scene, w, h, max_iterations=8, frame_lanes=2)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=8)
    for step in range(9):
        if step == 2:
            r.set_mode(False); o.set_mode(False)
        if step == 4:
            r.set_mode(True); o.set_mode(True)
        if step == 6:
            r.set_max_iterations(3); o.set_max_iterations(3)
        if step == 7:
            r.set_max_iterations(11); o.set_max_iterations(11)
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts()), step
        assert np.array_equal(r.accumulator(), o.accumulator()), step
    assert np.array_equal(r.pixels(), o.pixels())
    assert r.guard_timeouts() == 0
    r.close()


def test_full_size_one_sample_per_tick_lanes_equal_one_population():
    """1920x1080, one sample per tick (the reference's mode): the automatic choice of a free-running context is two lanes;
    1, 2 (free-running) and 4 (ordered strictly on the caller's stream) lanes give the same accumulator, display and counters."""
    scene = ptss.Scene("mixed")
    w, h, bounces = 1920, 1080, 8
    out = {}
    for lanes in (1, 0, 4):
        r = ptss.Renderer(scene, w, h, max_iterations=bounces, sync_each_frame=False, frame_lanes=lanes, lanes_free_run=(lanes == 0))
        if lanes == 0:
            assert r.frame_lanes == 2
        for _ in range(3):
            r.generate_frame()
        out[lanes] = (r.accumulator(), r.pixels(), r.live_counts().copy(), r.total_ray_bounces())
        assert r.guard_timeouts() == 0
        r.close()
    for lanes in (0, 4):
        assert np.array_equal(out[1][2], out[lanes][2]) and out[1][3] == out[lanes][3]
        assert np.array_equal(out[1][0], out[lanes][0])
        assert np.array_equal(out[1][1], out[lanes][1])


def test_automatic_lane_count():
    """Ordered strictly on the caller's stream (the default) a second lane gains nothing, so the library picks one; only a
    context that opted into free-running lanes (cfg.lanesFreeRun) gets the size rule."""
    scene = ptss.Scene("cornell")
    for w, h, S, free_run, want in ((64, 64, 1, True, 1),        # 4,096 rays per pass: one launch round, nothing to overlap
                                    (1280, 720, 1, True, 2),     # 0.9 million: ten launches of 1-2 resident rounds each
                                    (1280, 720, 1, False, 1),    # ... but not without the opt-in
                                    (1920, 1080, 1, False, 1),
                                    (640, 480, 1, True, 1),      # 0.3 million: a pass is ten launch latencies
                                    (1920, 1080, 40, True, 1)):  # 83 million rays per pass: launches are wide enough
        r = ptss.Renderer(scene, w, h, samples_per_pass=S, lanes_free_run=free_run)
        assert r.frame_lanes == want, (w, h, S, free_run)
        r.close()


def test_lanes_inside_pixel_band_shards():
    """Frame lanes and the multi-GPU pixel-band sharding compose: three shard contexts with two lanes each reassemble to
    the oracle's frame (at a size where the sharded loop guard never differs)."""
    import tiles
    w, h, bounces, world, band = 96, 54, 6, 3, 4
    scene = ptss.Scene("mixed")
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces)
    rs = [ptss.Renderer(scene, w, h, max_iterations=bounces, tile_rank=k, tile_world=world, band_rows=band, frame_lanes=2,
                        float_accumulator=True) for k in range(world)]
    for _ in range(3):
        o.generate_frame()
        for r in rs:
            r.generate_frame()
    assert np.array_equal(tiles.untile([r.accumulator() for r in rs], w, h, band), o.accumulator())
    assert np.array_equal(tiles.untile([r.pixels() for r in rs], w, h, band), o.pixels())
    assert np.array_equal(tiles.untile([r.float_accumulator() for r in rs], w, h, band), o.float_sum(), equal_nan=True)
    assert sum(r.total_ray_bounces() for r in rs) == o.total_ray_bounces()
    for r in rs:
        assert r.guard_timeouts() == 0
        r.close()


def test_lanes_with_callers_stream_and_buffers():
    """The caller's stream is the join point: after generate_frame + a synchronize of THAT stream alone the caller-owned
    accumulator and display buffer hold the frame, although the lanes ran on streams of their own."""
    import torch
    scene = ptss.Scene("cornell")
    w = h = 96
    r = ptss.Renderer(scene, w, h, max_iterations=5, sync_each_frame=False, frame_lanes=2)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=5)
    acc = torch.zeros((w * h, 3), dtype=torch.int32, device="cuda")
    pix = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    r.bind_accumulator(acc.data_ptr())
    s = torch.cuda.Stream()
    r.set_stream(s.cuda_stream)
    for _ in range(4):
        r.generate_frame(pix.data_ptr())
        o.generate_frame()
    s.synchronize()                                   # the caller's stream only
    with torch.cuda.stream(s):
        a, p = acc.cpu().numpy(), pix.cpu().numpy()
    assert np.array_equal(a.astype(np.uint32), o.accumulator())
    assert np.array_equal(p, o.pixels())
    r.close()


@pytest.mark.parametrize("S,ticks", [(1, 600), (5, 150)])
def test_lanes_free_running_for_many_frames(S, ticks):
    """Lanes are forked from the caller's stream only at a reset and may drift up to a frame apart: hundreds of ticks without a
    synchronisation in between (and a camera move in the middle = a reset and a fork) must end on the one-lane image. With
    S > 1 the pass's samples are added up by displayKernel on the CALLER's stream, behind the join, while the lanes are already
    parking the next frame's: the sample words are double-buffered by frame parity (round 3; before, samples could be lost)."""
    scene = ptss.Scene("mixed")
    w, h, bounces = 640, 360, 8
    out = {}
    for lanes in (1, 2, 3):
        r = ptss.Renderer(scene, w, h, max_iterations=bounces, sync_each_frame=False, frame_lanes=lanes, lanes_free_run=True, samples_per_pass=S)
        cam = ptss.default_camera()
        for t in range(ticks):
            if t == ticks * 5 // 12:
                cam.position.x = 0.125
                r.set_camera(cam)
            r.generate_frame()
        out[lanes] = (r.accumulator(), r.pixels(), r.total_ray_bounces(), r.live_counts().copy())
        assert r.guard_timeouts() == 0
        r.close()
    for lanes in (2, 3):
        assert out[1][2] == out[lanes][2] and np.array_equal(out[1][3], out[lanes][3])
        assert np.array_equal(out[1][0], out[lanes][0])
        assert np.array_equal(out[1][1], out[lanes][1])


@pytest.mark.parametrize("lanes,free_run", [(2, False), (3, False), (1, False)])
def test_work_enqueued_between_frames_is_ordered_like_on_one_stream(lanes, free_run):
    """STRICT lane ordering (the default): what the caller enqueues on its stream between two ptss_generate_frame calls sees
    exactly the frames before it, and the next frame's kernels wait for it — although the lanes run on streams of their own.
    Between the frames the caller's stream here runs a long filler (so that the next call is issued while it is still busy),
    snapshots the display buffer and the bound accumulator, and zero-fills the accumulator WITHOUT a reset: every snapshot
    must then hold exactly one frame's samples, frame by frame as the oracle produces them."""
    import torch
    scene = ptss.Scene("mixed")
    w, h, bounces, frames = 320, 180, 6, 6
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, sync_each_frame=False, frame_lanes=lanes, lanes_free_run=free_run)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces)
    acc = torch.zeros((w * h, 3), dtype=torch.int32, device="cuda")
    pix = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    filler = torch.ones((64 << 20,), dtype=torch.float32, device="cuda")   # 256 MB: ~0.1 ms per pass over it
    torch.cuda.synchronize()
    r.bind_accumulator(acc.data_ptr())
    s = torch.cuda.Stream()
    r.set_stream(s.cuda_stream)
    snaps_acc, snaps_pix = [], []
    with torch.cuda.stream(s):
        for _ in range(frames):
            r.generate_frame(pix.data_ptr())
            for _ in range(8):
                filler.mul_(1.0000001)             # keeps the caller's stream busy while the next frame is being enqueued
            snaps_acc.append(acc.clone())
            snaps_pix.append(pix.clone())
            acc.zero_()                            # frame k + 1 must add into zeros, not be zeroed half-way
    s.synchronize()
    r.synchronize()
    prev = np.zeros((w * h, 3), dtype=np.uint32)
    for k in range(frames):
        o.generate_frame()
        total = o.accumulator()
        assert np.array_equal(snaps_acc[k].cpu().numpy().astype(np.uint32), total - prev), k
        prev = total.copy()
    # the display value of frame k divides the accumulator the kernel saw (one frame's samples) by k + 1: only frame 0's
    # equals the oracle's; what must hold for every k is that the snapshot is the frame's own, complete display
    assert np.array_equal(snaps_pix[0].cpu().numpy(), ptss_display_of(snaps_acc[0].cpu().numpy(), 1))
    for k in range(frames):
        assert np.array_equal(snaps_pix[k].cpu().numpy(), ptss_display_of(snaps_acc[k].cpu().numpy(), k + 1)), k
    assert r.guard_timeouts() == 0
    r.close()
    o.close()


def ptss_display_of(acc, samples):
    """writeToPixelsKernel's display value (CudaTracer.cu:94-98): uchar(total * (1.f / samples) + 0.5f), w = 255."""
    inv = np.float32(1.0) / np.float32(samples)
    rgb = (acc.astype(np.float32) * inv + np.float32(0.5)).astype(np.uint32).astype(np.uint8)
    return np.concatenate([rgb, np.full((acc.shape[0], 1), 255, dtype=np.uint8)], axis=1)


def test_a_healthy_multi_lane_run_reports_no_timeout():
    """Waits between lanes are bounded (about two seconds); an expired wait would surface as PTSS_ETIMEOUT from the next
    synchronising call. A healthy run returns PTSS_OK everywhere and the counter reads 0."""
    import ctypes as C
    scene = ptss.Scene("cornell")
    r = ptss.Renderer(scene, 40, 20, max_iterations=15, frame_lanes=4)     # tiny: every lane asks its peers at some bounce
    L = ptss.device_lib()
    for _ in range(5):
        assert L.ptss_generate_frame(r._ctx, None, r.ticks) == 0            # syncEachFrame: checks the counter itself
        r.ticks += 1
    assert L.ptss_synchronize(r._ctx) == 0
    assert r.guard_timeouts() == 0
    r.close()
