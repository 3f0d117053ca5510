"""Host-only pieces of the drop-in: Camera()/moveCamera (CudaTracer.cu:822-870), the TGA writer
(saveScreenshot, :795-813) and the pixel-tile row map used for multi-GPU sharding."""
import math
import os

import numpy as np
import pytest

import ptss
import tiles


def test_camera_defaults():
    c = ptss.default_camera()  # RenderStructs.h:51-52
    assert (c.rotation.w, c.rotation.x, c.rotation.y, c.rotation.z) == (1, 0, 0, 0)
    assert c.position.tuple() == (0, 0, 0)
    assert c.zNear == np.float32(-0.1) and c.zFar == -100 and c.fieldOfView == np.float32(math.pi) / np.float32(2)


def test_move_camera_keys():
    c = ptss.default_camera()
    assert ptss.move_camera(c, "w") and np.allclose(c.position.tuple(), (0, 0, -0.2))
    assert ptss.move_camera(c, "d") and np.allclose(c.position.tuple(), (0.2, 0, -0.2))
    assert ptss.move_camera(c, "q") and np.allclose(c.position.tuple(), (0.2, 0.2, -0.2))
    for k in "ase":
        assert ptss.move_camera(c, k)
    assert np.allclose(c.position.tuple(), (0, 0, 0), atol=1e-7)
    assert not ptss.move_camera(c, "x") and not ptss.move_camera(c, " ")
    # 'f' = yaw +10 degrees about +Y; 36 of them come back to the start; forward then points along -x
    assert ptss.move_camera(c, "f")
    assert abs(c.rotation.y - math.sin(math.radians(5))) < 1e-6 and abs(c.rotation.w - math.cos(math.radians(5))) < 1e-6
    for _ in range(8):
        ptss.move_camera(c, "f")       # 90 degrees
    ptss.move_camera(c, "w")
    assert np.allclose(c.position.tuple(), (-0.2, 0, 0), atol=1e-6)
    for _ in range(27):
        ptss.move_camera(c, "f")
    q = np.array([c.rotation.w, c.rotation.x, c.rotation.y, c.rotation.z])
    assert abs(abs(q[0]) - 1) < 1e-5 and abs(np.linalg.norm(q) - 1) < 1e-6
    c = ptss.default_camera()
    ptss.move_camera(c, "t")
    assert c.rotation.x > 0        # pitch up
    ptss.move_camera(c, "g")
    assert abs(c.rotation.x) < 1e-7


def test_tga_writer(tmp_path):
    h, w = 3, 5
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = np.arange(w)[None, :] * 10          # R
    img[..., 1] = np.arange(h)[:, None] * 20          # G
    img[..., 2] = 7                                   # B
    img[..., 3] = 255
    p = str(tmp_path / "shot.tga")
    ptss.write_tga(p, img)
    b = open(p, "rb").read()
    assert len(b) == 18 + w * h * 3
    assert list(b[:18]) == [0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w, 0, h, 0, 24, 0]   # CudaTracer.cu:804
    body = np.frombuffer(b, np.uint8, offset=18).reshape(h, w, 3)
    assert np.array_equal(body[..., 0], img[..., 2]) and np.array_equal(body[..., 2], img[..., 0])  # BGR
    big = np.zeros((2, 300, 4), np.uint8)
    ptss.write_tga(p, big)
    assert open(p, "rb").read()[12:16] == bytes([300 % 256, 300 // 256, 2, 0])
    with pytest.raises(ptss.PtssError):
        ptss.write_tga(str(tmp_path / "no" / "dir.tga"), img)


@pytest.mark.parametrize("height,band,world", [(1080, 8, 1), (1080, 8, 2), (1080, 8, 8), (90, 8, 4), (7, 8, 3), (64, 1, 5)])
def test_tile_rows_partition_the_frame(height, band, world):
    rows = tiles.rank_rows(height, band, world)
    allr = np.concatenate(rows)
    assert sorted(allr.tolist()) == list(range(height))          # every row exactly once
    for r, rr in enumerate(rows):
        assert all(((y // band) % world) == r for y in rr) and list(rr) == sorted(rr)
    if height % (band * world) == 0:
        assert len({len(r) for r in rows}) == 1                  # balanced when it divides


def test_untile_roundtrip():
    w, h, band, world = 12, 37, 4, 3
    frame = np.arange(w * h * 3, dtype=np.uint32).reshape(w * h, 3)
    parts = [tiles.extract(frame, w, h, band, r, world) for r in range(world)]
    assert sum(len(p) for p in parts) == w * h
    assert np.array_equal(tiles.untile(parts, w, h, band), frame)
