"""The closest-hit triangle loop picks its loop body per triangle from the triangle's EDGE CLASSES (csrc/pttri.h): an edge the
host has found to run along one coordinate axis (two exact-zero components) loses the products with those zeros. The claim,
checked here on the host build of that very header (ptss_probe_triangle_forms, libptss_host.so; float32, -ffp-contract=off):

  for finite origins, directions and vertices the class form accepts exactly the triangles the general form
  (Triangle::intersectRay, Primitives.h:25-83) accepts, with bit-identical distance and bit-identical NON-ZERO weights; a weight
  that is exactly zero may carry the other sign — the kernel re-evaluates such a hit with the general form (closestHit,
  `zeroWeight`) — and nothing else differs.

The cases: the presets' own triangles (all 13 class codes occur) against random, axis-parallel, grazing and edge-on rays;
sign-of-zero corners (direction or origin components of +-0, origins on the triangle's plane and on its edges, hits exactly on
an edge and on a vertex); products that overflow or go subnormal; and the precondition itself — a non-finite origin makes
the forms differ, which is why the kernel tests |o|^2 once per query. CPU only; the GPU parity suite runs the kernels."""
import ctypes as C

import numpy as np
import pytest

import ptss

AXES = np.eye(3, dtype=np.float32)


def forms(tri, o, d, limit=None, primary=False):
    tri = np.ascontiguousarray(tri, dtype=np.float32).reshape(-1, 9)
    n = tri.shape[0]
    o = np.ascontiguousarray(np.broadcast_to(np.asarray(o, dtype=np.float32), (n, 3)))
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(d, dtype=np.float32), (n, 3)))
    lim = np.full(n, np.inf, dtype=np.float32) if limit is None else np.ascontiguousarray(np.broadcast_to(np.asarray(limit, dtype=np.float32), (n,)))
    cls = np.zeros(n, dtype=np.int32)
    g = np.zeros((n, 6), dtype=np.float32)
    c = np.zeros((n, 6), dtype=np.float32)
    f32p = C.POINTER(C.c_float)
    rc = ptss.host_lib().ptss_probe_triangle_forms(tri.ctypes.data_as(f32p), o.ctypes.data_as(f32p), d.ctypes.data_as(f32p),
                                                   lim.ctypes.data_as(f32p), 1 if primary else 0, n,
                                                   cls.ctypes.data_as(C.POINTER(C.c_int)), g.ctypes.data_as(f32p), c.ctypes.data_as(f32p))
    assert rc == 0
    return cls, g, c


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def assert_same_up_to_zero_signs(g, c, what=""):
    """accepted alike; for accepted hits: same distance bits, weights bit-identical unless exactly zero (then zero in both)."""
    assert np.array_equal(g[:, 0], c[:, 0]), what
    hit = g[:, 0] == 1.0
    assert np.array_equal(bits(g[hit, 1]), bits(c[hit, 1])), what            # distance: accepted hits have dist > 0
    for col in (2, 3, 4):
        a, b = g[hit, col], c[hit, col]
        nz = (a != 0) | (b != 0)
        assert np.array_equal(bits(a[nz]), bits(b[nz])), (what, col)
        assert np.all(a[~nz] == 0) and np.all(b[~nz] == 0)
    # the determinant of EVERY pair (accepted or not) agrees whenever it is non-zero, and so does the distance wherever the
    # reference looks at it at all (|det| > 1e-7, Primitives.h:41; a determinant of +-0 makes 1 / det an infinity of either sign)
    det_g, det_c = g[:, 5], c[:, 5]
    assert np.array_equal(np.isnan(det_g), np.isnan(det_c)), what
    nz = ~np.isnan(det_g) & ((det_g != 0) | (det_c != 0))
    assert np.array_equal(bits(det_g[nz]), bits(det_c[nz])), what
    with np.errstate(invalid="ignore"):
        used = np.abs(det_g) > np.float32(1e-7)
    a, b = g[used, 1], c[used, 1]
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    nz = ~np.isnan(a) & ((a != 0) | (b != 0))
    assert np.array_equal(bits(a[nz]), bits(b[nz])), what
    return int(hit.sum())


def preset_triangles():
    rows = []
    for preset in ("mixed", "cornell", "default"):
        d = ptss.Scene(preset).desc
        for i in range(d.numTriangles):
            t = d.triangles[i]
            v = [np.array([p.x, p.y, p.z], dtype=np.float32) for p in (t.vertex0, t.vertex1, t.vertex2)]
            rows.append(np.concatenate([v[0], v[1] - v[0], v[2] - v[0]]))
    return np.array(rows, dtype=np.float32)


def synthetic_triangles(rng, n):
    """every class code: edges along axes (either sign, any length), mixed with general edges"""
    rows = []
    for _ in range(n):
        v0 = rng.uniform(-6, 6, 3)
        edges = []
        for _e in range(2):
            k = rng.integers(0, 4)
            if k == 3:
                edges.append(rng.uniform(-8, 8, 3))
            else:
                e = np.zeros(3)
                e[k] = rng.uniform(0.1, 9) * rng.choice([-1, 1])
                e[e == 0] = rng.choice([0.0, -0.0], size=2)   # either sign of zero
                edges.append(e)
        rows.append(np.concatenate([v0, edges[0], edges[1]]))
    return np.array(rows, dtype=np.float32)


def test_every_class_code_occurs_in_the_presets():
    tri = preset_triangles()
    cls, _, _ = forms(tri, [0, 0, 0], [0, 0, -1])
    # 12 of the 16 triangles of the default / mixed box have two axis-parallel edges, the 88-degree wall and the front wall one
    mixed = cls[:16]
    assert sorted(set(mixed.tolist())) == [4 * a + b for a in (1, 2, 3) for b in (0, 1, 2, 3) if a != b and (a, b) not in ((1, 0), (3, 0))] or len(set(mixed.tolist())) >= 8
    assert np.sum((mixed // 4 != 0) & (mixed % 4 != 0)) == 12 and np.sum(mixed == 0) == 0
    rng = np.random.default_rng(7)
    syn, _, _ = forms(synthetic_triangles(rng, 4000), [0, 0, 0], [0, 0, -1])
    assert set(syn.tolist()) == {4 * a + b for a in range(4) for b in range(4)} - {5, 10, 15}


@pytest.mark.parametrize("primary", [False, True])
def test_random_rays_against_preset_and_synthetic_triangles(primary):
    rng = np.random.default_rng(20261004)
    tris = np.concatenate([preset_triangles(), synthetic_triangles(rng, 600)])
    total = 0
    for _ in range(60):
        k = rng.integers(0, len(tris), size=4096)
        o = rng.uniform(-5, 5, (4096, 3)).astype(np.float32)
        d = rng.normal(size=(4096, 3)).astype(np.float32)
        t = tris[k]
        uv = rng.uniform(-0.3, 1.0, (4096, 2)).astype(np.float32)   # three rays in four are aimed at (or just past) their triangle
        aim = (t[:, 0:3] + uv[:, :1] * t[:, 3:6] + uv[:, 1:] * t[:, 6:9]) - o
        d = np.where(rng.random((4096, 1)) < 0.75, aim, d).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        limit = np.where(rng.random(4096) < 0.5, np.inf, rng.uniform(0.5, 12, 4096)).astype(np.float32)
        _, g, c = forms(t, o, d, limit, primary)
        total += assert_same_up_to_zero_signs(g, c, "random")
    assert total > 10000   # plenty of accepted hits among them


@pytest.mark.parametrize("primary", [False, True])
def test_sign_of_zero_corners(primary):
    """Directions and origins with exact zeros of either sign (axis-parallel rays, rays inside the triangle's plane, origins on
    the plane, on an edge, on a vertex) against every class; hits exactly on an edge (a weight of exactly zero) included."""
    rng = np.random.default_rng(11)
    tris = np.concatenate([preset_triangles(), synthetic_triangles(rng, 300)])
    zeros = [0.0, -0.0]
    dirs = []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            for z1 in zeros:
                for z2 in zeros:
                    v = [z1, z2]
                    v.insert(ax, sgn)
                    dirs.append(v)
    for a in range(3):             # diagonal in a coordinate plane, third component +-0
        for z in zeros:
            for s1 in (1, -1):
                for s2 in (1, -1):
                    v = [s1 * 0.70710678, s2 * 0.70710678]
                    v.insert(a, z)
                    dirs.append(v)
    dirs = np.array(dirs, dtype=np.float32)
    n_hits = n_zero_w = 0
    backs = np.array([0.0, 1.0, 2.5], dtype=np.float32)   # origin on the point itself, or stepped back along the ray (exact in float for axis rays)
    for t in tris:
        v0, e1, e2 = t[0:3], t[3:6], t[6:9]
        # points of interest on the triangle: vertices, edge midpoints, the centroid, a point on each edge's extension
        pts = np.array([v0, v0 + e1, v0 + e2, v0 + 0.5 * e1, v0 + 0.5 * e2, v0 + 0.5 * e1 + 0.5 * e2, v0 + 0.25 * e1 + 0.25 * e2, v0 - 0.5 * e1],
                       dtype=np.float32)
        o = (pts[:, None, None, :] - backs[None, None, :, None] * dirs[None, :, None, :]).astype(np.float32).reshape(-1, 3)
        dd = np.broadcast_to(dirs[None, :, None, :], (len(pts), len(dirs), len(backs), 3)).reshape(-1, 3)
        _, g, c = forms(np.broadcast_to(t, (len(o), 9)), o, dd, None, primary)
        n_hits += assert_same_up_to_zero_signs(g, c, t)
        n_zero_w += int(np.sum((g[:, 0] == 1.0) & ((g[:, 3] == 0) | (g[:, 4] == 0))))
    assert n_hits > 1000 and n_zero_w > 50    # edge-on hits with an exactly-zero weight did occur and were accepted alike


def test_large_and_tiny_products_inside_the_domain():
    """The kernel's domain (closestHit): every |coordinate| <= 1e15 (SceneLayout::triClassed, geometryBounded), |d|^2 < 2^30,
    |o|^2 < 2^100. There d x e2 and (o - v0) x e1 stay finite (< 1e32), so every product with a flagged component is an exact
    zero; final products may still overflow or go subnormal — in both forms alike."""
    rng = np.random.default_rng(5)
    tris = synthetic_triangles(rng, 400)
    for scale_t, scale_o, scale_d in ((5e13, 1e14, 1.0), (5e13, 1.0, 5e3), (1e-20, 1e-20, 1.0), (1.0, 1.0, 1e-30), (5e13, 1e14, 1e-10),
                                      (1e-30, 1e-30, 1e-8), (1e-5, 1e14, 5e3), (5e13, 1e-3, 1e-38)):
        k = rng.integers(0, len(tris), size=8192)
        o = (rng.uniform(-5, 5, (8192, 3)) * scale_o).astype(np.float32)
        d = (rng.normal(size=(8192, 3)) * scale_d).astype(np.float32)
        t = (tris[k].astype(np.float64) * scale_t).astype(np.float32)
        assert np.abs(t[:, :3]).max() <= 1e15 and np.abs(t[:, :3] + t[:, 3:6]).max() <= 1e15 and np.abs(t[:, :3] + t[:, 6:9]).max() <= 1e15
        assert (d.astype(np.float64) ** 2).sum(axis=1).max() < 2.0 ** 30 and (o.astype(np.float64) ** 2).sum(axis=1).max() < 2.0 ** 100
        _, g, c = forms(t, o, d)
        assert_same_up_to_zero_signs(g, c, (scale_t, scale_o, scale_d))


def test_the_domain_matters():
    """Outside it the forms DO differ. A non-finite origin: the general form meets inf * 0 = NaN where the class form has left
    the product out — which is why closestHit tests |o|^2 < 2^100 (and |d|^2 < 2^30) once per query before it takes the class
    bodies. Coordinates near the top of the float range: d x e2 itself overflows, and the general form's e1.y * (d x e2).y =
    0 * inf is NaN — which is why the host records classes only for bounded geometry (every |coordinate| <= 1e15)."""
    t = np.array([[-1, -1, -5, 2, 0, 0, 0, 2, 0]], dtype=np.float32)     # e1 along x, e2 along y: class 1 * 4 + 2
    cls, g, c = forms(t, [np.inf, 0.2, 0], [0, 0, -1])
    assert cls[0] == 6
    assert np.isnan(g[0, 1]) and not np.isnan(c[0, 1])
    # ... while any finite origin of the domain keeps them together
    _, g, c = forms(t, [1e15, 0.2, 0], [0, 0, -1])
    assert_same_up_to_zero_signs(g, c)
    big = np.array([[1e37, 1e37, 1e37, 3e38, 0, 0, 3e38, 0, 3e38]], dtype=np.float32)   # e1 along x; e2 in the x-z plane, huge
    cls, g, c = forms(big, [0, 0, 0], [0.6, 0, -0.8])      # (d x e2).y = d.z e2.x - d.x e2.z overflows
    assert cls[0] == 4 and np.isnan(g[0, 5]) and not np.isnan(c[0, 5])
