"""The oracle against the committed golden vectors (tests/golden/, made by make_golden.py): guards
the oracle, the scene builder and the shared math against silent drift on any host."""
import json
import os

import numpy as np
import pytest

import oracle
import ptss

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run(preset, g, upto=None):
    w, h, bounces, ticks, seed = [int(x) for x in g["meta"][:5]]
    S = int(g["meta"][5]) if len(g["meta"]) > 5 else 1
    ticks = upto or ticks
    scene = ptss.Scene(preset)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, seed=seed, samples_per_pass=S)
    live = []
    rad0 = None
    for t in range(ticks):
        o.generate_frame()
        live.append(o.live_counts())
        if t == 0:
            rad0 = o.last_radiance0()
    return o, np.array(live, dtype=np.uint32), rad0


def test_xorwow_vectors():
    want = json.load(open(os.path.join(GOLDEN, "xorwow.json")))
    for key, v in want.items():
        seed, sub = [int(x) for x in key.split(":")]
        st, raw, uni = oracle.probe_rng(seed, sub, 16)
        assert st.tolist() == v["state"] and raw.tolist() == v["raw"]
        assert [np.float32(u).tobytes().hex() for u in uni] == v["uniform_hex"]


def test_c1_cornell_full():
    g = np.load(os.path.join(GOLDEN, "c1_cornell.npz"))
    o, live, rad0 = _run("cornell", g)
    assert np.array_equal(live, g["live_counts"])
    assert np.array_equal(rad0, g["radiance0_tick0"], equal_nan=True)
    assert np.array_equal(o.accumulator(), g["accumulator"].astype(np.uint32))
    assert np.array_equal(o.pixels()[:, :3], g["pixels"])
    assert (o.pixels()[:, 3] == 255).all()
    assert o.total_ray_bounces() == int(g["total_ray_bounces"][0])


def test_c1_default_first_ticks():
    g = np.load(os.path.join(GOLDEN, "c1_default.npz"))
    o, live, _ = _run("default", g, upto=3)
    assert np.array_equal(live, g["live_counts"][:3])


def test_small_mixed_full():
    g = np.load(os.path.join(GOLDEN, "small_mixed.npz"))
    o, live, rad0 = _run("mixed", g)
    assert np.array_equal(live, g["live_counts"])
    assert np.array_equal(o.accumulator(), g["accumulator"].astype(np.uint32))
    assert np.array_equal(o.float_sum(), g["float_sum"], equal_nan=True)
    assert np.array_equal(rad0, g["radiance0_tick0"], equal_nan=True)


@pytest.mark.parametrize("name,preset", [("small_stress", "stress"), ("small_mixed_s4", "mixed")])
def test_many_spheres_and_sample_lanes(name, preset):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    o, live, _ = _run(preset, g)
    assert np.array_equal(live, g["live_counts"])
    assert np.array_equal(o.accumulator(), g["accumulator"].astype(np.uint32))
    assert np.array_equal(o.pixels()[:, :3], g["pixels"])
    assert np.array_equal(o.float_sum(), g["float_sum"], equal_nan=True)
    assert o.total_ray_bounces() == int(g["total_ray_bounces"][0])


def test_oracle_semantics_ticks_reset_and_mode():
    scene = ptss.Scene("cornell")
    o = oracle.Oracle(scene.desc, 32, 32, max_iterations=3)
    o.generate_frame(ticks=1)
    o.generate_frame(ticks=2)
    a2 = o.accumulator()
    assert a2.max() <= 2 * 255
    o.request_reset()                      # resetTicksThisFrame: clear, lastResetTick = ticks (CudaTracer.cu:602-608)
    o.generate_frame(ticks=3)
    assert o.accumulator().max() <= 255
    o.set_mode(False)                      # ray tracing: one bounce, which is the last one (CudaTracer.cu:620)
    o.generate_frame(ticks=4)
    assert len(o.live_counts()) == 1 and o.live_counts()[0] == 32 * 32
    # RNG streams are never re-seeded (CudaTracer.cu:724 only): a reset does not replay samples
    assert not np.array_equal(o.accumulator(), a2)


def test_literal_slot_rng_probe_differs_only_statistically():
    # SURVEY.md §9.2 fidelity probe: slot-bound RNG + numRays/96 truncation changes samples, not the image
    scene = ptss.Scene("cornell")
    a = oracle.Oracle(scene.desc, 48, 48, max_iterations=4)
    b = oracle.Oracle(scene.desc, 48, 48, max_iterations=4, literal_slot_rng=True)
    for _ in range(24):
        a.generate_frame()
        b.generate_frame()
    ma, mb = a.accumulator().mean(0) / 24, b.accumulator().mean(0) / 24
    assert np.abs(ma - mb).max() < 6.0     # same mean image brightness to within noise
    assert b.live_counts()[0] == (48 * 48 // 96) * 96


def test_samples_per_pass_extension_semantics():
    """S sample lanes per pixel (extension; S = 1 is the reference): lane l of pixel p owns subsequence p*S + l,
    every sample is tone-mapped on its own, the display divides by S*(ticks+1)."""
    scene = ptss.Scene("cornell")
    w = h = 24
    a = oracle.Oracle(scene.desc, w, h, max_iterations=4, samples_per_pass=1)
    b = oracle.Oracle(scene.desc, w, h, max_iterations=4, samples_per_pass=4)
    assert np.array_equal(b.rng_state(5, 0), oracle.probe_rng(0x5EED, 5 * 4 + 0, 0)[0])
    assert np.array_equal(b.rng_state(5, 3), oracle.probe_rng(0x5EED, 5 * 4 + 3, 0)[0])
    assert np.array_equal(a.rng_state(5), oracle.probe_rng(0x5EED, 5, 0)[0])
    for _ in range(8):
        a.generate_frame()
    for _ in range(2):
        b.generate_frame()
    assert b.live_counts()[0] == 4 * w * h
    acc_a, acc_b = a.accumulator(), b.accumulator()           # 8 samples per pixel each, different streams
    assert acc_b.max() <= 255 * 8 and not np.array_equal(acc_a, acc_b)
    assert abs(acc_a.mean() - acc_b.mean()) / acc_a.mean() < 0.05
    px = b.pixels()
    assert np.array_equal(px[:, :3], (acc_b * np.float32(1.0) / np.float32(8) + np.float32(0.5)).astype(np.uint8)) or \
        np.array_equal(px[:, :3], (acc_b * (np.float32(1.0) / np.float32(8)) + np.float32(0.5)).astype(np.uint8))
