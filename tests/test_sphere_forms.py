"""The chunked traversal decides "this sphere may be hit" (Primitives.h:115-118: discriminent = b*b - 4*c with b = 2 d.v, rejected
when < 0) from h = d.v and c as  !(h*h < c)  instead of  !((2h)*(2h) < 4*c)  (shiftInSphereBounded, ptss_kernels.hip). The two
verdicts are the same for every float32 h whenever c is zero, NaN, or 2^-105 <= |c| <= 2^105 — what accelEligible
(ptss_api.hip) guarantees for c = |v|^2 - r^2 — and for infinite c together with a non-finite h (a non-finite origin).
Checked here in float32 arithmetic on the adversarial corners and on random bit patterns. CPU only; no product code runs."""
import numpy as np


def reference_form(h, c):
    with np.errstate(all="ignore"):
        b = np.float32(2) * h
        return ~((b * b) < (np.float32(4) * c))


def bounded_form(h, c):
    with np.errstate(all="ignore"):
        return ~((h * h) < c)


def floats(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def eligible_c(c):
    a = np.abs(c)
    return np.isnan(c) | (c == 0) | ((a >= np.float32(2.0 ** -105)) & (a <= np.float32(2.0 ** 105)))


def test_corner_operands():
    tiny = [0.0, 1e-45, 2.0 ** -149, 2.0 ** -140, 2.0 ** -127, 2.0 ** -126, 2.0 ** -64, 2.0 ** -63, 2.0 ** -62, 1e-12, 1.0, 3.0,
            2.0 ** 52, 2.0 ** 62, 2.0 ** 63, 1.5 * 2.0 ** 63, 2.0 ** 64 * 0.999, 2.0 ** 100, 3.4e38, np.inf, np.nan]
    hs = np.array([s * v for v in tiny for s in (1.0, -1.0)], dtype=np.float32)
    cs = np.array([s * v for v in [0.0, 2.0 ** -105, 2.0 ** -104, 1e-24, 2.0 ** -52, 1e-6, 1.0, 2.0 ** 52, 2.0 ** 104, 2.0 ** 105, np.nan]
                   for s in (1.0, -1.0)], dtype=np.float32)
    H, C = np.meshgrid(hs, cs)
    assert eligible_c(C).all()
    assert np.array_equal(reference_form(H, C), bounded_form(H, C))
    # an infinite c comes from a non-finite origin only, and then h is not finite either
    for c in (np.float32(np.inf), np.float32(-np.inf)):
        for h in (np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)):
            assert reference_form(h, c) == bounded_form(h, c)


def test_random_bit_patterns():
    rng = np.random.default_rng(20260412)
    for _ in range(8):
        h = floats(rng.integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64))
        c = floats(rng.integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64))
        keep = eligible_c(c)
        assert keep.sum() > 100000
        assert np.array_equal(reference_form(h[keep], c[keep]), bounded_form(h[keep], c[keep]))
        # near-ties: c within a few ulps of h*h (where a wrong rounding argument would show first)
        hh = h[np.isfinite(h)]
        with np.errstate(all="ignore"):
            sq = hh * hh
        for k in (-2, -1, 0, 1, 2):
            cc = (sq.view(np.uint32).astype(np.int64) + k).clip(0, 2 ** 32 - 1).astype(np.uint32).view(np.float32)
            keep = eligible_c(cc)
            assert np.array_equal(reference_form(hh[keep], cc[keep]), bounded_form(hh[keep], cc[keep]))


def test_the_domain_matters():
    """Outside the domain the forms DO differ (why accelEligible bounds the geometry): a subnormal h*h against a subnormal c —
    4 h*h keeps two bits that h*h has already lost."""
    h, c = floats([0x1CF9A8CB])[0], floats([0x0000079C])[0]
    assert not eligible_c(np.array([c], dtype=np.float32))[0]
    assert reference_form(h, c) != bounded_form(h, c)
