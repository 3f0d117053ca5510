"""Pixel-tile sharding on real hardware: N contexts on one GPU, each owning the interleaved row
bands of one rank, must together reproduce the unsharded frame bit for bit (the property the
multi-GPU run relies on), and at BASELINE's full 1920x1080 size the path is checked through
size-independent properties instead of the (too slow) oracle."""
import numpy as np
import pytest

import oracle
import ptss
import tiles

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,band,S", [(2, 8, 1), (3, 4, 1), (8, 8, 1), (4, 8, 4)])
def test_tiles_reassemble_to_the_oracle_frame(world, band, S):
    w, h, bounces, spp = 64, 45, 5, 4
    scene = ptss.Scene("mixed")
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S)
    rs = [ptss.Renderer(scene, w, h, max_iterations=bounces, tile_rank=k, tile_world=world, band_rows=band,
                        float_accumulator=True, samples_per_pass=S) for k in range(world)]
    for _ in range(spp):
        o.generate_frame()
        for r in rs:
            r.generate_frame()
    assert sum(r.local_pixels for r in rs) == w * h
    for k, r in enumerate(rs):
        assert np.array_equal(r.rows(), ptss.tile_rows(h, band, k, world))
    acc = tiles.untile([r.accumulator() for r in rs], w, h, band)
    pix = tiles.untile([r.pixels() for r in rs], w, h, band)
    fsum = tiles.untile([r.float_accumulator() for r in rs], w, h, band)
    assert np.array_equal(acc, o.accumulator())
    assert np.array_equal(pix, o.pixels())
    assert np.array_equal(fsum, o.float_sum(), equal_nan=True)
    assert sum(r.total_ray_bounces() for r in rs) == o.total_ray_bounces()
    for r in rs:
        r.close()


def test_full_size_properties_1080p():
    """configs[2]/[3] at full size: determinism, tile invariance, counter consistency, accumulator bounds."""
    w, h, bounces, spp, band = 1920, 1080, 8, 3, 8
    scene = ptss.Scene("mixed")
    a = ptss.Renderer(scene, w, h, max_iterations=bounces)
    b = ptss.Renderer(scene, w, h, max_iterations=bounces, sync_each_frame=False)
    parts = [ptss.Renderer(scene, w, h, max_iterations=bounces, tile_rank=k, tile_world=2, band_rows=band) for k in range(2)]
    live_sum = 0
    for _ in range(spp):
        a.generate_frame()
        live = a.live_counts()
        assert live[0] == w * h and (np.diff(live.astype(np.int64)) <= 0).all()   # compaction only shrinks the set
        live_sum += int(live.sum())
        b.generate_frame()
        for p in parts:
            p.generate_frame()
    acc = a.accumulator()
    assert a.total_ray_bounces() == live_sum                     # device counter == sum of per-bounce live counts
    assert np.array_equal(acc, b.accumulator())                  # same seed, async vs sync driver: identical
    assert acc.max() <= 255 * spp and acc.sum() > 0
    whole = tiles.untile([p.accumulator() for p in parts], w, h, band)
    assert np.array_equal(whole, acc)                            # 2-way shard == unsharded, every pixel
    px = a.pixels()
    assert (px[:, 3] == 255).all()
    assert np.array_equal(px[:, :3], ((acc * np.float32(1.0 / spp)) + np.float32(0.5)).astype(np.uint8))
    for r in [a, b] + parts:
        r.close()


def test_loop_guard_is_a_whole_frame_quantity_only_unsharded():
    """The one documented difference (include/ptss.h): the reference's `numRays > 128` guard
    (CudaTracer.cu:622) is evaluated on the frame's live count; a shard cannot know it and never stops early.
    16x8 = 128 pixels: unsharded, not even bounce 0 runs (the oracle agrees: every sample is radiance 0); two shards of 64
    pixels each trace their rays to the end."""
    w, h, bounces = 16, 8, 4
    scene = ptss.Scene("cornell")
    whole = ptss.Renderer(scene, w, h, max_iterations=bounces)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces)
    parts = [ptss.Renderer(scene, w, h, max_iterations=bounces, tile_rank=k, tile_world=2, band_rows=4) for k in range(2)]
    for _ in range(2):
        whole.generate_frame()
        o.generate_frame()
        for p in parts:
            p.generate_frame()
    assert np.array_equal(whole.accumulator(), o.accumulator()) and whole.accumulator().max() == 0
    assert whole.live_counts().sum() == 0 and whole.total_ray_bounces() == 0
    for p in parts:
        assert p.live_counts()[0] == 64                     # its own rays, although 64 <= 128
    sharded = tiles.untile([p.accumulator() for p in parts], w, h, 4)
    assert sharded.max() > 0 and sum(p.total_ray_bounces() for p in parts) > 128
    for r in [whole] + parts:
        r.close()
