"""The one artefact the reference holds for this path: CudaTracer/image.tga (512x512, the render README.md:32 describes;
Scene.cpp's Cornell-box alternative with the four defined spheres = this repo's `cornell` preset).

QUALITATIVE ONLY. The image comes from an older, brighter revision of the reference (no mirror panel by the right wall,
~2000 spp, unknown seed), so it pins no number: parity stays "unpinned by the reference" (DESIGN.md §4). What it can show
is that the oracle's frame has the reference's structure — same framing, the red wall on the left, the green one on the
right, the light patch in the ceiling, the darkest object where the reference has it — which a wrong camera model, a
flipped axis, a wrong scene table or a wrong row order would break. Reads /root/reference (this container only; skipped
where it does not exist, e.g. on the GPU box)."""
import os

import numpy as np
import pytest

import oracle
import ptss

TGA = "/root/reference/CudaTracer/image.tga"
pytestmark = pytest.mark.skipif(not os.path.exists(TGA), reason="the reference tree is not present here")


def read_tga(path):
    raw = open(path, "rb").read()
    h18 = raw[:18]
    assert h18[2] == 2 and h18[16] == 24, "uncompressed 24-bit true-colour TGA (saveScreenshot, CudaTracer.cu:795-813)"
    w, h = h18[12] | h18[13] << 8, h18[14] | h18[15] << 8
    bgr = np.frombuffer(raw[18:18 + w * h * 3], dtype=np.uint8).reshape(h, w, 3)   # row 0 = bottom (GL order)
    return bgr[:, :, ::-1].astype(np.float64)


def blocks(img, b=64):
    h, w, _ = img.shape
    return img.reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))[::-1]   # flip: block row 0 = top of the picture


def test_cornell_preset_has_the_structure_of_the_reference_image():
    ref = read_tga(TGA)
    assert ref.shape == (512, 512, 3)
    scene = ptss.Scene("cornell")
    o = oracle.Oracle(scene.desc, 512, 512, max_iterations=15, seed=0x5EED)   # CudaUtils.h:7 DIM, CudaTracer.h:39
    spp = 8
    for _ in range(spp):
        o.generate_frame()
    ours = o.accumulator().astype(np.float64).reshape(512, 512, 3) / spp        # row 0 = bottom, like the TGA
    o.close()
    R, O = blocks(ref), blocks(ours)
    for B in (R, O):
        left = B[2:6, 0:2].mean(axis=(0, 1))        # the left wall: red, nothing else
        assert left[0] > 100 and left[0] > 10 * max(left[1], left[2], 1.0), left
        right = B[2:5, 6].mean(axis=0)              # the right wall: green dominates (ours has a mirror panel beside it)
        assert right[1] > 1.5 * right[0] and right[1] > 1.5 * right[2], right
        light = B[0, 3:5].mean(axis=0)              # the ceiling light: saturated white, the brightest blocks of the frame
        lum = B.mean(axis=2)
        assert light.min() > 235 and lum.max() == pytest.approx(lum[0, 3:5].max(), abs=2.0), light
        inner = B[4:8, 2:6].mean(axis=2)            # the darkest object of the lower middle sits in the same block
        assert np.unravel_index(np.argmin(inner), inner.shape) == (2, 2), inner
    # same picture block by block, per channel (the reference is brighter; correlation ignores gain)
    for c in range(3):
        assert np.corrcoef(R[..., c].ravel(), O[..., c].ravel())[0, 1] > 0.75
