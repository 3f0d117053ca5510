"""Drop-in semantics of the frame callback on the GPU: ticks / reset / camera / ray-tracing mode
behave like generateFrame + Key + moveCamera of the reference, checked against the oracle driven
with the same calls."""
import numpy as np
import pytest

import oracle
import ptss

pytestmark = pytest.mark.gpu


def _pair(preset="cornell", w=48, h=48, bounces=4):
    scene = ptss.Scene(preset)
    return scene, ptss.Renderer(scene, w, h, max_iterations=bounces), oracle.Oracle(scene.desc, w, h, max_iterations=bounces)


def _same(r, o):
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.live_counts(), o.live_counts())


def test_ticks_need_not_start_at_one_and_reset_restarts_the_average():
    scene, r, o = _pair()
    for t in (7, 8, 9):                      # GPUAnimBitmap's counter is whatever it is; lastResetTick anchors it
        r.generate_frame(ticks=t)
        o.generate_frame(ticks=t)
    _same(r, o)
    r.request_reset()
    o.request_reset()
    r.generate_frame(ticks=10)
    o.generate_frame(ticks=10)
    _same(r, o)
    assert r.accumulator().max() <= 255       # one sample since the reset
    r.close()


def test_camera_moves_like_the_reference_keys():
    scene, r, o = _pair()
    r.generate_frame(); o.generate_frame()
    cam = r.get_camera()
    for key in "wwdtf":                       # forward x2, right, pitch up, yaw left (CudaTracer.cu:822-870)
        assert ptss.move_camera(cam, key)
    r.set_camera(cam)                         # sets resetTicksThisFrame, as Key() does (:782-785)
    o.set_camera(cam)
    for _ in range(3):
        r.generate_frame(); o.generate_frame()
    _same(r, o)
    got = r.get_camera()
    assert (got.position.x, got.position.y, got.position.z) == (cam.position.x, cam.position.y, cam.position.z)
    r.close()


def test_ray_tracing_mode_is_one_bounce():
    scene, r, o = _pair()
    r.set_mode(False); o.set_mode(False)      # space bar (:760-765)
    for _ in range(2):
        r.generate_frame(); o.generate_frame()
    assert len(r.live_counts()) == 1
    _same(r, o)
    r.set_mode(True); o.set_mode(True)
    r.set_max_iterations(6); o.set_max_iterations(6)
    r.generate_frame(); o.generate_frame()
    assert len(r.live_counts()) == 6
    _same(r, o)
    r.close()


def test_no_display_buffer_and_external_buffers():
    import torch
    scene = ptss.Scene("cornell")
    w = h = 32
    r = ptss.Renderer(scene, w, h, max_iterations=3, sync_each_frame=False)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=3)
    acc = torch.zeros((w * h, 3), dtype=torch.int32, device="cuda")
    pix = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    r.bind_accumulator(acc.data_ptr())        # torch owns totalPixelColors
    torch.cuda.synchronize()                  # the zero-fills above ran on torch's default stream; `s` does not wait for it
    s = torch.cuda.Stream()
    r.set_stream(s.cuda_stream)
    for _ in range(3):
        r.generate_frame(pix.data_ptr())      # torch owns the "PBO"
        o.generate_frame()
    r.synchronize()
    s.synchronize()
    assert np.array_equal(acc.cpu().numpy().astype(np.uint32), o.accumulator())
    assert np.array_equal(pix.cpu().numpy(), o.pixels())
    r.close()


def test_last_pass_ms_is_reported_when_syncing():
    scene, r, o = _pair()
    r.generate_frame()
    assert 0.0 < r.last_pass_ms() < 1000.0    # the "Time per pass" of CudaTracer.cu:641-645
    r.close()


def test_errors_are_codes_not_exits():
    scene, r, o = _pair()
    with pytest.raises(ptss.PtssError):
        r.set_max_iterations(0)
    with pytest.raises(ptss.PtssError):
        r.rng_state(10 ** 9)
    with pytest.raises(ptss.PtssError):
        r.float_accumulator()                  # context was created without floatAccumulator
    with pytest.raises(ptss.PtssError):
        ptss.Renderer(scene, 32, 32, device=99)
    r.close()
