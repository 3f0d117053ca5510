"""The bounded sphere candidate test also drops spheres that lie BEHIND the ray's origin (ptss_kernels.hip aheadFactor /
shiftInSphere<true>): with h = d.v and c = |v|^2 - r^2 it keeps a sphere when  !(h * m < c),  m = min(h * 2^-18, h)  — that is
h*h >= c for h <= 0 (the discriminant test of Primitives.h:115-118, unchanged bits) and 2^-18 h*h >= c for h > 0. A dropped
sphere must be one the reference's whole test (Primitives.h:107-175) rejects for EVERY running distance: discriminant < 0, or
both roots negative. Checked here in float32 arithmetic on corner operands, random bit patterns and operands a few ulps around
the threshold, on the domain the bounded form is used on (tests/test_sphere_forms.py: c zero, NaN or 2^-105 <= |c| <= 2^105, as
ptss_create guarantees for bounded geometry; an infinite c only beside a non-finite h). CPU only; no product code runs."""
import numpy as np

K = np.float32(2.0 ** -18)


def floats(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def reference_can_accept(h, c):
    """Sphere::intersectRay with distance = +inf: True where the reference would take the hit (Primitives.h:109-174)."""
    with np.errstate(all="ignore"):
        b = np.float32(2) * h
        disc = b * b - np.float32(4) * c
        rejected = disc < 0
        s = np.sqrt(disc)
        t0 = (-b + s) * np.float32(0.5)
        t1 = (-b - s) * np.float32(0.5)
        rejected = rejected | ((t0 < 0) & (t1 < 0))
        swap = t0 > t1
        lo = np.where(swap, t1, t0)
        hi = np.where(swap, t0, t1)
        cand = np.where(lo < 0, hi, lo)
        rejected = rejected | (cand > np.float32(np.inf))   # never: `cand > distance` with distance = inf
    return ~rejected


def kept(h, c):
    with np.errstate(all="ignore"):
        hk = h * K
        m = np.fmin(hk, h)          # v_min_f32: the smaller; NaN only if both are
        m = np.where(np.isnan(h), h, m)
        return ~((h * m) < c)


def eligible(h, c):
    a = np.abs(c)
    return np.isnan(c) | (c == 0) | ((a >= np.float32(2.0 ** -105)) & (a <= np.float32(2.0 ** 105))) | (np.isinf(c) & ~np.isfinite(h))


def check(h, c):
    h = np.asarray(h, dtype=np.float32)
    c = np.asarray(c, dtype=np.float32)
    keep = eligible(h, c)
    h, c = h[keep], c[keep]
    acc = reference_can_accept(h, c)
    k = kept(h, c)
    bad = acc & ~k
    assert not bad.any(), (h[bad][:4], c[bad][:4])
    return acc, k


def test_ahead_of_the_origin_nothing_changes():
    """h <= 0: the kept set is the discriminant test's, bit for bit."""
    rng = np.random.default_rng(7)
    h = -np.abs(floats(rng.integers(0, 2 ** 32, size=1 << 18, dtype=np.uint64)))
    c = floats(rng.integers(0, 2 ** 32, size=1 << 18, dtype=np.uint64))
    with np.errstate(all="ignore"):
        assert np.array_equal(kept(h, c), ~((h * h) < c))


def test_corner_operands():
    mags = [0.0, 1e-45, 2.0 ** -140, 2.0 ** -126, 2.0 ** -100, 2.0 ** -63, 1e-12, 1e-7, 1e-4, 0.5, 1.0, 3.0, 1e4, 2.0 ** 52, 2.0 ** 63,
            2.0 ** 100, 3.4e38, np.inf, np.nan]
    hs = np.array([s * v for v in mags for s in (1.0, -1.0)], dtype=np.float32)
    cs = np.array([s * v for v in mags for s in (1.0, -1.0)], dtype=np.float32)
    H, C = np.meshgrid(hs, cs)
    check(H.ravel(), C.ravel())


def test_random_bit_patterns_and_the_threshold():
    rng = np.random.default_rng(20261005)
    dropped_behind = 0
    for _ in range(6):
        h = floats(rng.integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64))
        c = floats(rng.integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64))
        check(h, c)
        # operands of a scene's size, positive h (the sphere's centre lies behind the origin's plane)
        hp = np.abs(rng.normal(0, 3, size=1 << 20)).astype(np.float32)
        for scale in (1e-8, 1e-6, 2.0 ** -18, 1e-5, 1e-4, 1e-2, 1.0):
            cc = (hp.astype(np.float64) ** 2 * scale * rng.uniform(0.5, 2.0, size=hp.size)).astype(np.float32)
            acc, k = check(hp, cc)
            dropped_behind += int((~k).sum())
        # c a few ulps around 2^-18 h*h — where a wrong rounding argument would show first — and around h*h
        with np.errstate(all="ignore"):
            for t in (hp * hp * K, hp * (hp * K), hp * hp):
                for d in range(-3, 4):
                    cc = (t.view(np.uint32).astype(np.int64) + d).clip(0, 2 ** 32 - 1).astype(np.uint32).view(np.float32)
                    check(hp, cc)
    assert dropped_behind > 1000000   # the test does drop spheres (a bumped origin leaving a sphere: c ~ 2e-4 r, h ~ r)


def test_a_ray_leaving_the_sphere_it_stands_on():
    """The case the cull is for: origin bumped 1e-4 off a sphere of radius r along the normal, direction anywhere in the outer
    half space: the reference rejects (both roots negative) and the candidate test now drops it; a direction into the sphere is kept."""
    rng = np.random.default_rng(3)
    r = rng.uniform(0.05, 2.0, size=200000).astype(np.float32)
    cosv = rng.uniform(0.01, 1.0, size=r.size).astype(np.float32)     # d . n, leaving
    v = r + np.float32(1e-4)
    h = (cosv * v).astype(np.float32)
    c = (v * v - r * r).astype(np.float32)
    acc, k = check(h, c)
    assert not acc.any()
    assert (~k).mean() > 0.95
    acc, k = check(-h, c)      # heading in: a true hit unless it grazes past; every hit is kept
    assert acc.mean() > 0.9 and k[acc].all()
