"""Edge scenes, assembled from raw C records, HIP path vs oracle, bit for bit: empty and one-sided scenes, odd
primitive counts (the candidate-mask chunks and the triangle loop have no padding to hide behind), degenerate
geometry that drives the reciprocal / square-root fast paths out of their proven ranges (zero determinants, zero
radii, overflowing squares -> inf and NaN: the wave-uniform IEEE escapes must then give what the CPU gives), the
camera inside a refracting sphere, the bounce-count limits, and frame sizes that divide by nothing."""
import ctypes as C

import numpy as np
import pytest

import oracle
import ptss
from ptss_types import AreaLight, Material, PointLight, SceneDesc, Sphere, Triangle

pytestmark = pytest.mark.gpu

CREAM, RED, GREEN, EMIT, MIRROR, GLASS, COOK, PHONG = 8, 9, 10, 11, 12, 3, 0, 6  # material classes of the 'mixed' preset


def _set3(v, xyz):
    v.x, v.y, v.z = (float(t) for t in xyz)


def build(spheres=(), triangles=(), area=(), point=()):
    """spheres: (centre, radius, material); triangles: (v0, v1, v2, material); area: (power, firstTriangle);
    point: (position, power). Materials are the 14 of the 'mixed' preset."""
    base = ptss.Scene("mixed")
    mats = (Material * base.desc.numMaterials)(*base.materials)
    sph = (Sphere * max(1, len(spheres)))()
    for i, (c, r, m) in enumerate(spheres):
        _set3(sph[i].position, c)
        sph[i].radius, sph[i].materialIdx = float(r), m
    tri = (Triangle * max(1, len(triangles)))()
    for i, (a, b, c, m) in enumerate(triangles):
        a, b, c = (np.asarray(t, np.float32) for t in (a, b, c))
        n = np.cross(b - a, c - a).astype(np.float32)
        ln = np.float32(np.sqrt(np.dot(n, n)))
        n = n / ln if ln > 0 else np.asarray([0, 1, 0], np.float32)
        for dst, src in ((tri[i].vertex0, a), (tri[i].vertex1, b), (tri[i].vertex2, c), (tri[i].normal0, n),
                         (tri[i].normal1, n), (tri[i].normal2, n)):
            _set3(dst, src)
        tri[i].materialIdx = m
    al = (AreaLight * max(1, len(area)))()
    for i, (p, first) in enumerate(area):
        _set3(al[i].power, p)
        al[i].area, al[i].triangleIdx, al[i].numTriangles = 1.0, first, 2
    pl = (PointLight * max(1, len(point)))()
    for i, (pos, p) in enumerate(point):
        _set3(pl[i].position, pos)
        _set3(pl[i].power, p)
    d = SceneDesc()
    d.spheres, d.numSpheres = (sph if spheres else None), len(spheres)
    d.triangles, d.numTriangles = (tri if triangles else None), len(triangles)
    d.materials, d.numMaterials = mats, len(mats)
    d.areaLights, d.numAreaLights = (al if area else None), len(area)
    d.pointLights, d.numPointLights = (pl if point else None), len(point)

    class Holder:
        pass
    h = Holder()
    h.desc, h.keep = d, (sph, tri, mats, al, pl, base)
    return h


def quad(p0, p1, p2, p3, m):
    return [(p0, p1, p2, m), (p0, p2, p3, m)]


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)) or np.array_equal(a, b, equal_nan=True)


def check(scene, w, h, bounces, ticks=2, S=1, nan_ok=False):
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, samples_per_pass=S)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S)
    for _ in range(ticks):
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts())
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    f, g = r.float_accumulator(), o.float_sum()
    assert np.array_equal(f, g, equal_nan=True) if nan_ok else np.array_equal(f, g)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    for p in (0, w * h - 1):
        assert np.array_equal(r.rng_state(p), o.rng_state(p, 0))
    counts = r.live_counts()
    r.close()
    return counts


FLOOR = quad((-4, -1, 0), (4, -1, 0), (4, -1, -9), (-4, -1, -9), CREAM)
LAMP = quad((-1, 3, -3), (1, 3, -3), (1, 3, -5), (-1, 3, -5), EMIT)


def test_empty_scene_every_ray_misses():
    counts = check(build(), 37, 19, 5)
    assert counts[0] == 37 * 19 and all(c == 0 for c in counts[1:])


def test_spheres_only_odd_count_across_a_mask_chunk_with_a_point_light():
    sph = [((-3.2 + 0.8 * (i % 9), -0.8 + 0.8 * (i // 9), -4.0 - 0.3 * (i % 4)), 0.33, (CREAM, MIRROR, GLASS, COOK, PHONG)[i % 5])
           for i in range(33)]
    counts = check(build(spheres=sph, point=[((0, 3, -3), (40, 40, 40))]), 48, 32, 6)
    assert counts[1] > 0


def test_single_sphere_and_single_triangle():
    check(build(spheres=[((0, 0, -3), 1.0, RED)], point=[((2, 2, 0), (30, 30, 30))]), 24, 24, 3)
    check(build(triangles=[((-2, -1, -3), (2, -1, -3), (0, 2, -3), GREEN)], point=[((0, 0, 0), (9, 9, 9))]), 24, 24, 3)


def test_triangles_only_odd_count_with_an_area_light():
    tris = LAMP + [((-3, -1, -2), (3, -1, -2), (0, -1, -8), CREAM)]       # light triangles first: triangleIdx 0 and 1
    counts = check(build(triangles=tris, area=[((60, 60, 60), 0)]), 40, 30, 5)
    assert counts[1] > 0


def test_no_lights_at_all():
    check(build(spheres=[((0, 0, -3), 1.0, CREAM)], triangles=FLOOR), 32, 20, 4)


def test_degenerate_primitives_take_the_ieee_escapes():
    """Zero-area triangles (det == 0 -> 1/det = inf through the IEEE escape, then rejected), a zero-radius sphere, a
    refracting sphere around the camera (every path starts on the inside branch with Beer-Lambert), a point light
    buried inside an opaque sphere (always occluded)."""
    tris = FLOOR + LAMP + [((1, 0, -3), (1, 0, -3), (1, 0, -3), RED), ((0, 0, -2), (0, 0, -2), (1, 1, -2), GREEN)]
    sph = [((0, 0, 0), 0.5, GLASS), ((1.5, 0, -4), 0.0, RED), ((-1.5, 0, -4), 0.7, CREAM), ((2.5, 2.0, -5), 0.4, MIRROR)]
    scene = build(spheres=sph, triangles=tris, area=[((50, 50, 50), 2)], point=[((-1.5, 0, -4), (25, 25, 25))])
    counts = check(scene, 48, 32, 6, ticks=3)
    assert counts[1] > 0


def test_overflowing_geometry_follows_the_same_inf_nan_path():
    """A sphere 1e20 away with radius 1e19: v.v and r^2 overflow to inf, the discriminant is inf - inf = NaN, and the
    reference's ordered compares decide what a NaN does (`!(disc < 0)` keeps it as a candidate). Integer results must
    still agree exactly; float sums may hold NaN, compared as equal-NaN."""
    sph = [((0, 0, -1e20), 1e19, CREAM), ((0, 0, -3), 0.8, COOK)]
    check(build(spheres=sph, triangles=FLOOR + LAMP, area=[((50, 50, 50), 2)]), 32, 20, 4, nan_ok=True)


def test_sphere_test_forms_switch_with_the_geometry_and_the_camera():
    """Scenes of finite, moderate geometry run the sphere candidate tests in their shorter form (SceneLayout::sphereBounded,
    tests/test_sphere_forms.py) as long as the camera is in range as well; a radius below 1e-12 or a coordinate beyond 1e15
    keeps the reference's literal form, and a camera that leaves the range switches instantiation for that frame. All of
    them against the oracle, bit for bit."""
    base = [((0, 0, -3), 0.8, COOK), ((-1.6, -0.2, -4), 0.7, GLASS), ((1.5, 0.1, -3.5), 0.6, MIRROR)]
    lit = dict(triangles=FLOOR + LAMP, area=[((50, 50, 50), 2)])
    check(build(spheres=base, **lit), 40, 24, 5)                                        # bounded
    check(build(spheres=base + [((0.5, -0.5, -2), 1e-13, RED)], **lit), 40, 24, 5)      # a radius under the bound
    check(build(spheres=base + [((0, 3e15, -3), 1.0, RED)], **lit), 40, 24, 5)          # a coordinate over the bound
    # the camera leaves the range and comes back: same context, three instantiations of the same frame loop
    scene = build(spheres=base, **lit)
    r = ptss.Renderer(scene, 40, 24, max_iterations=5, float_accumulator=True)
    o = oracle.Oracle(scene.desc, 40, 24, max_iterations=5)
    cam = ptss.default_camera()
    for z in (0.0, 4e15, 0.5):
        cam.position.z = z
        r.set_camera(cam)
        o.set_camera(cam)
        for _ in range(2):
            r.generate_frame()
            o.generate_frame()
            assert np.array_equal(r.live_counts(), o.live_counts()), z
        assert np.array_equal(r.accumulator(), o.accumulator()), z
        assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True), z
    r.close()


def test_diffuse_scenes_pair_their_shadow_segments():
    """Scenes whose primitives are (at least four in five) diffuse, with two or more lights, run the kernels that carry the two
    shadow segments of a surface point as ONE queue entry and test them together (SceneLayout::neePairs, pairAnyHit). Three
    lights: the second round holds a single segment per entry; a point light inside the scene; lanes whose one light is
    below the horizon. Against the oracle, bit for bit, dense passes (full frames) and split ones (tiny frames)."""
    sph = [((-1.8, -0.3, -4.2), 0.7, CREAM), ((0.1, -0.4, -3.4), 0.6, RED), ((1.7, -0.2, -4.6), 0.8, GREEN), ((0.2, 0.9, -5.5), 0.5, CREAM),
           ((-0.9, 1.6, -4.8), 0.4, GREEN), ((1.1, 1.4, -3.9), 0.35, RED), ((0, -0.7, -2.2), 0.3, CREAM)]
    scene = build(spheres=sph, triangles=FLOOR + LAMP, area=[((50, 50, 50), 2)],
                  point=[((-2.5, 2.0, -2.0), (30, 30, 30)), ((0.3, 0.2, -4.4), (6, 6, 6))])
    check(scene, 64, 40, 6, ticks=3)
    check(scene, 17, 9, 5, ticks=2, S=3)
    two = build(spheres=sph, triangles=FLOOR + LAMP, area=[((50, 50, 50), 2)], point=[((-2.5, 2.0, -2.0), (30, 30, 30))])
    check(two, 48, 32, 8, ticks=2, S=2)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_diffuse_scenes_with_several_lights(seed):
    """Random all-diffuse scenes (the paired-shadow kernels, SceneLayout::neePairs) with 2-5 lights, odd and even counts,
    lights inside the cloud of spheres, triangles at random: against the oracle, at sizes that give dense and split passes."""
    rng = np.random.default_rng(seed)
    mats = [CREAM, RED, GREEN]
    sph = [((float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-0.9, 2.2)), float(rng.uniform(-7.0, -2.5))), float(rng.uniform(0.15, 0.7)),
            mats[int(rng.integers(0, 3))]) for _ in range(int(rng.integers(5, 30)))]
    tris = FLOOR + LAMP
    for _ in range(int(rng.integers(0, 6))):
        a = rng.uniform(-3, 3, size=3) + np.array([0, 0.5, -5])
        tris = tris + [(tuple(a), tuple(a + rng.uniform(-1.5, 1.5, size=3)), tuple(a + rng.uniform(-1.5, 1.5, size=3)), mats[int(rng.integers(0, 3))])]
    pts = [((float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.5, 2.8)), float(rng.uniform(-6.5, -1.5))), tuple(float(x) for x in rng.uniform(5, 40, size=3)))
           for _ in range(int(rng.integers(1, 5)))]
    scene = build(spheres=sph, triangles=tris, area=[((50, 50, 50), 2)], point=pts)
    check(scene, int(rng.integers(40, 90)), int(rng.integers(24, 60)), int(rng.integers(2, 9)), ticks=2, S=int(rng.choice([1, 2, 5])))
    check(scene, 13, 7, 4, ticks=2)


def test_walls_made_of_giant_spheres():
    """The smallpt way of building a room: walls are spheres of radius 1e5. Leaving such a wall, c = |v|^2 - r^2 is ~2 r bump =
    20 against b^2 ~ 4e10 — below half an ulp of b^2, so the discriminant rounds to b^2, one root comes out as exactly 0 and
    the reference's `t0 < 0 && t1 < 0` exit does not fire (Primitives.h:137): whatever the literal arithmetic does there,
    both sphere-test forms (geometry is bounded: the shorter one runs) must do the same, bit for bit."""
    R = 1e5
    sph = [((-R - 3, 0, -5), R, RED), ((R + 3, 0, -5), R, GREEN), ((0, 0, -R - 9), R, CREAM), ((0, -R - 2, -5), R, CREAM),
           ((0, R + 3, -5), R, CREAM), ((-1.2, -1.2, -6), 0.8, MIRROR), ((1.1, -1.3, -4.5), 0.7, GLASS), ((0, -0.5, -5.5), 0.4, COOK)]
    counts = check(build(spheres=sph, point=[((0, 2.5, -5), (60, 60, 60))]), 56, 40, 8, ticks=3)
    assert counts[3] > 0
    check(build(spheres=sph, point=[((0, 2.5, -5), (60, 60, 60))]), 40, 24, 6, ticks=2, S=3)


@pytest.mark.parametrize("bounces", [1, 64])
def test_bounce_count_limits_in_a_mirror_box(bounces):
    box = (quad((-2, -2, 1), (2, -2, 1), (2, -2, -6), (-2, -2, -6), MIRROR) + quad((-2, 2, 1), (-2, 2, -6), (2, 2, -6), (2, 2, 1), MIRROR) +
           quad((-2, -2, 1), (-2, -2, -6), (-2, 2, -6), (-2, 2, 1), MIRROR) + quad((2, -2, 1), (2, 2, 1), (2, 2, -6), (2, -2, -6), MIRROR) +
           quad((-2, -2, -6), (2, -2, -6), (2, 2, -6), (-2, 2, -6), CREAM) + quad((-2, -2, 1), (-2, 2, 1), (2, 2, 1), (2, -2, 1), MIRROR) +
           quad((-0.5, 1.99, -2), (0.5, 1.99, -2), (0.5, 1.99, -3), (-0.5, 1.99, -3), EMIT))
    counts = check(build(triangles=box, area=[((30, 30, 30), 12)]), 20, 16, bounces)
    assert len(counts) == bounces


@pytest.mark.parametrize("w,h,S", [(97, 53, 1), (1, 1, 1), (3, 1, 4), (131, 7, 2)])
def test_frame_sizes_that_divide_by_nothing(w, h, S):
    check(ptss.Scene("mixed"), w, h, 3, ticks=2, S=S)


def test_triangles_hit_at_exactly_the_same_distance_tie_like_the_reference():
    """Bounded scenes store their triangles grouped by edge class (pttri.h) and visit them class by class; the reference's
    sequential rule `dist <= distance` (Primitives.h:52) lets the LAST of several triangles hit at exactly the same distance win.
    The closest hit keeps the key (distance, 0xFFFFFFFE - original index) instead. Exact ties made on purpose: every wall
    triangle of a box appears two or three times with different materials — as an exact duplicate (same class, same arithmetic),
    with its second and third vertex swapped (class (a, b) becomes (b, a): other products, the same plane), and rotated
    (v1, v2, v0: general edges) — interleaved so that the winner by index is now an earlier, now a later stored position; a
    sphere touching a wall and a light panel lying IN the ceiling's plane add sphere/triangle and triangle/triangle ties."""
    X, Y, Z0, Z1 = 3.0, 2.5, 0.5, -7.0
    walls = []
    mats = [CREAM, RED, GREEN, MIRROR, PHONG, COOK]
    corners = {
        "floor": ((-X, -Y, Z0), (X, -Y, Z0), (X, -Y, Z1), (-X, -Y, Z1)),
        "ceil": ((-X, Y, Z0), (-X, Y, Z1), (X, Y, Z1), (X, Y, Z0)),
        "back": ((-X, -Y, Z1), (X, -Y, Z1), (X, Y, Z1), (-X, Y, Z1)),
        "left": ((-X, -Y, Z0), (-X, -Y, Z1), (-X, Y, Z1), (-X, Y, Z0)),
        "right": ((X, -Y, Z0), (X, Y, Z0), (X, Y, Z1), (X, -Y, Z1)),
    }
    tris = []
    for k, (name, (p0, p1, p2, p3)) in enumerate(corners.items()):
        for (a, b, c) in ((p0, p1, p2), (p0, p2, p3)):
            tris.append((a, b, c, mats[k % len(mats)]))                 # the wall itself
            tris.append((a, c, b, mats[(k + 1) % len(mats)]))           # second and third vertex swapped: the other class, the other facing
            tris.append((a, b, c, mats[(k + 2) % len(mats)]))           # exact duplicate, later index
            if k % 2 == 0:
                tris.append((b, c, a, mats[(k + 3) % len(mats)]))       # rotated: one general edge
    light = len(tris)
    tris += quad((-1, Y, -2), (-1, Y, -4), (1, Y, -4), (1, Y, -2), EMIT)  # IN the ceiling's plane
    tris += [(t[0], t[1], t[2], CREAM) for t in tris[:4]]                # the first wall once more, at the very end
    spheres = [((0.0, -Y + 1.0, -4.0), 1.0, GLASS), ((-X + 0.75, 0.0, -3.0), 0.75, COOK), ((1.5, 0.5, -5.0), 0.8, MIRROR)]
    s = build(spheres=spheres, triangles=tris, area=[((60, 60, 50), light)])
    check(s, 96, 64, 6, ticks=3)
    check(s, 64, 48, 5, ticks=2, S=3)
