"""The oracle built a second time WITHOUT the product's math header (oracle/libm_math.h: libm float functions, plain vector
arithmetic; -DORACLE_LIBM_MATH) against the bit-exact checker that shares csrc/ptmath.h with the kernels.

Why: every GPU parity test is an array_equal against an oracle that includes the product's own ptm::sin/atan/log/exp/pow and
fma chains — a mistake there would be wrong on both sides. The two builds here share no arithmetic. They differ in the last
ulp, so a few percent of the samples take another branch somewhere and individual pixels differ; what must agree is the
image statistically. Parity stays "unpinned by the reference" (it holds no fixtures): this pins the shared
math against libm, not against CUDA."""
import numpy as np
import pytest

import oracle
import ptss


def _render(preset, w, h, bounces, spp, math):
    scene = ptss.Scene(preset)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, seed=0x5EED, math=math)
    first = None
    for k in range(spp):
        o.generate_frame()
        if k == 0:
            first = o.accumulator().copy()
    acc = o.accumulator().astype(np.float64) / spp
    o.close()
    return first, acc.reshape(h, w, 3)


def _blocks(img, b):
    h, w, _ = img.shape
    return img[: h // b * b, : w // b * b].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("preset,w,h,bounces,spp", [("cornell", 96, 96, 4, 48), ("mixed", 128, 72, 8, 32), ("pointlight", 64, 64, 5, 32)])
def test_libm_build_agrees_statistically(preset, w, h, bounces, spp):
    f_a, a = _render(preset, w, h, bounces, spp, "ptmath")
    f_b, b = _render(preset, w, h, bounces, spp, "libm")
    # one sample per pixel: almost every pixel takes the same decisions in both builds
    same = (f_a == f_b).all(axis=1).mean()
    assert same > 0.9, same
    # the accumulated images: same brightness per channel to 1.5 %, same structure block by block
    ma, mb = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mb) <= 0.015 * np.maximum(ma, 1.0)), (ma, mb)
    ba, bb = _blocks(a, 8), _blocks(b, 8)
    assert np.corrcoef(ba.ravel(), bb.ravel())[0, 1] > 0.995
    assert np.abs(ba - bb).mean() < 4.0   # of 255, per 8x8 block: Monte-Carlo noise of the differing samples


def test_libm_build_really_is_another_arithmetic():
    """Guards the guard: if both libraries were the same build the test above would prove nothing."""
    _, a = _render("mixed", 64, 36, 8, 8, "ptmath")
    _, b = _render("mixed", 64, 36, 8, 8, "libm")
    assert not np.array_equal(a, b)
