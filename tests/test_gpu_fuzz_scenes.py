"""Randomised scenes, HIP path vs oracle, bit for bit. Every record is drawn at random — material classes with arbitrary
diffuse / specular / refractive weights (also summing above 1 and to 0), finite, huge and infinite Phong exponents,
indices of refraction on both sides of 1, roughness from mirror-like to matte, all four values of the two flag bits,
absorbing media, emissive surfaces; spheres that overlap, nest and contain the camera; sliver and huge triangles with
unnormalised vertex normals; point lights and area lights, also none at all — so the comparison is not limited to the
material classes and layouts the presets happen to use. Camera moves and a mode toggle ride along."""
import ctypes as C

import numpy as np
import pytest

import oracle
import ptss
from ptss_types import AreaLight, Material, PointLight, SceneDesc, Sphere, Triangle

pytestmark = pytest.mark.gpu


def _set3(v, xyz):
    v.x, v.y, v.z = (float(t) for t in xyz)


def random_scene(seed, ns=None, nt=None):
    rng = np.random.default_rng(seed)
    nm = int(rng.integers(2, 9))
    mats = (Material * nm)()
    for m in mats:
        kind = rng.integers(0, 12)
        _set3(m.diffuseColor, rng.random(3))
        _set3(m.specularColor, rng.random(3))
        _set3(m.absorption, rng.random(3) * rng.choice([0.0, 0.3, 3.0]))
        _set3(m.emmitance, rng.random(3) * rng.choice([0.0, 0.0, 0.0, 2.0]))
        m.specularExponent = float(rng.choice([0.0, 1.0, 7.5, 250.0, 1e6, np.inf]))
        m.indexOfRefraction = float(rng.choice([0.6, 1.0, 1.33, 1.55, 2.5, 5.8]))
        w = rng.random(3) * rng.choice([0.0, 0.5, 1.0, 1.6], size=3)   # weights may sum above 1 or vanish
        m.diffAvg, m.specAvg, m.refrAvg = (float(t) for t in w)
        if kind == 0:
            m.diffAvg, m.specAvg, m.refrAvg = 0.7, 0.0, 0.0        # plain Lambert
        elif kind == 1:
            m.diffAvg, m.specAvg, m.refrAvg = 0.0, 0.0, 0.0        # absorbs everything
        m.roughness = float(rng.choice([0.0, 0.02, 0.3, 0.9]))
        m.flags = bytes([int(rng.choice([0, 0, 1, 2, 3]))])
    ns0, nt0 = int(rng.integers(0, 40)), int(rng.integers(0, 24))
    ns, nt = (ns0 if ns is None else ns), (nt0 if nt is None else nt)
    sph = (Sphere * max(ns, 1))()
    for i in range(ns):
        _set3(sph[i].position, rng.uniform(-3, 3, 3) + np.array([0, 0, -4.0]))
        sph[i].radius = float(rng.choice([0.05, 0.3, 0.8, 2.5]) * rng.uniform(0.5, 1.5))
        sph[i].materialIdx = int(rng.integers(0, nm))
    if ns and rng.random() < 0.3:                                    # the camera starts inside this one
        _set3(sph[0].position, (0.1, -0.1, 0.2))
        sph[0].radius = 0.9
    tri = (Triangle * max(nt, 2))()
    for i in range(nt):
        base = rng.uniform(-4, 4, 3) + np.array([0, 0, -5.0])
        scale = rng.choice([0.01, 1.0, 6.0])
        _set3(tri[i].vertex0, base)
        _set3(tri[i].vertex1, base + rng.normal(0, scale, 3))
        _set3(tri[i].vertex2, base + rng.normal(0, scale, 3))
        for n in (tri[i].normal0, tri[i].normal1, tri[i].normal2):
            _set3(n, rng.normal(0, 1, 3))                             # not normalised, not even consistent: as given
        tri[i].materialIdx = int(rng.integers(0, nm))
    na = int(rng.integers(0, 3)) if nt >= 2 else 0
    al = (AreaLight * max(na, 1))()
    for i in range(na):
        _set3(al[i].power, rng.random(3) * 80)
        al[i].area, al[i].triangleIdx, al[i].numTriangles = 1.0, int(rng.integers(0, nt - 1)), 2
    npnt = int(rng.integers(0, 4))
    pl = (PointLight * max(npnt, 1))()
    for i in range(npnt):
        _set3(pl[i].position, rng.uniform(-3, 3, 3) + np.array([0, 1, -3.0]))
        _set3(pl[i].power, rng.random(3) * 60)
    d = SceneDesc()
    d.spheres, d.numSpheres = (sph if ns else None), ns
    d.triangles, d.numTriangles = (tri if nt else None), nt
    d.materials, d.numMaterials = mats, nm
    d.areaLights, d.numAreaLights = (al if na else None), na
    d.pointLights, d.numPointLights = (pl if npnt else None), npnt
    _set3(d.defaultColor, rng.random(3) * rng.choice([0.0, 1.0]))

    class Holder:
        pass
    h = Holder()
    h.desc, h.keep = d, (sph, tri, mats, al, pl)
    return h, rng


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_matches_oracle(seed):
    scene, rng = random_scene(1000 + seed)
    w, h = int(rng.integers(40, 120)), int(rng.integers(24, 80))
    bounces, S = int(rng.integers(1, 12)), int(rng.choice([1, 1, 2, 5]))
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, samples_per_pass=S, seed=seed * 7919 + 1)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S, seed=seed * 7919 + 1)
    cam = ptss.default_camera()
    for tick in range(4):
        if tick == 2:                                                # move the camera (resets the average) ...
            for key in rng.choice(list("wasdqezxcv"), size=3):        # the reference's movement keys, CudaTracer.cu:822-870
                ptss.move_camera(cam, str(key))
            r.set_camera(cam)
            o.set_camera(cam)
        if tick == 3 and seed % 3 == 0:                              # ... or drop to one-bounce ray tracing
            r.set_mode(False)
            o.set_mode(False)
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts()), (seed, tick)
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    r.close()


def axis_scene(seed):
    """Triangles whose edges are drawn from all the edge classes of pttri.h (along x, y or z; in a coordinate plane; general),
    on a quarter-unit grid so that edges are shared, triangles coincide (exact distance ties) and rays graze edges and
    corners — the cases in which the class forms of the kernels must give the general form's bits. Materials and spheres as
    in random_scene; area lights sit on class triangles."""
    scene, rng = random_scene(seed, nt=0)
    d = scene.desc
    nt = int(rng.integers(6, 40))
    tri = (Triangle * nt)()

    def edge(cls):
        e = np.zeros(3)
        if cls in (1, 2, 3):
            e[cls - 1] = rng.choice([-1, 1]) * rng.integers(1, 17) * 0.25
        elif cls == 4:                                               # in a coordinate plane: one exact zero (class 0 in the kernels)
            e = rng.integers(-8, 9, 3) * 0.25
            e[rng.integers(0, 3)] = 0.0
        else:
            e = rng.normal(0, 1.5, 3)
        if rng.random() < 0.2:
            e = np.where(e == 0.0, -0.0, e)                            # negative zeros are zeros too
        return e
    i = 0
    while i < nt:
        base = rng.integers(-12, 13, 3) * 0.25 + np.array([0, 0, -4.0])
        e1, e2 = edge(int(rng.integers(0, 6))), edge(int(rng.integers(0, 6)))
        copies = 2 if (rng.random() < 0.25 and i + 1 < nt) else 1    # the same triangle twice: an exact tie, lowest original index wins
        for _ in range(copies):
            _set3(tri[i].vertex0, base)
            _set3(tri[i].vertex1, base + e1)
            _set3(tri[i].vertex2, base + e2)
            for n in (tri[i].normal0, tri[i].normal1, tri[i].normal2):
                _set3(n, rng.normal(0, 1, 3))
            tri[i].materialIdx = int(rng.integers(0, d.numMaterials))
            i += 1
    na = int(rng.integers(0, 3))
    al = (AreaLight * max(na, 1))()
    for k in range(na):
        _set3(al[k].power, rng.random(3) * 80)
        al[k].area, al[k].triangleIdx, al[k].numTriangles = 1.0, int(rng.integers(0, nt - 1)), 2
    d.triangles, d.numTriangles = tri, nt
    d.areaLights, d.numAreaLights = (al if na else None), na
    scene.keep = scene.keep + (tri, al)
    return scene, rng


@pytest.mark.parametrize("seed", range(10))
def test_axis_aligned_triangle_scene_matches_oracle(seed):
    scene, rng = axis_scene(5000 + seed)
    w, h = int(rng.integers(48, 128)), int(rng.integers(32, 80))
    bounces, S = int(rng.integers(2, 10)), int(rng.choice([1, 2, 4]))
    r = ptss.Renderer(scene, w, h, max_iterations=bounces, float_accumulator=True, samples_per_pass=S, seed=seed * 104729 + 3)
    o = oracle.Oracle(scene.desc, w, h, max_iterations=bounces, samples_per_pass=S, seed=seed * 104729 + 3)
    cam = ptss.default_camera()
    for tick in range(3):
        if tick == 1:                                                # an axis-parallel view from a grid point: rays along edges and through corners
            cam.position.x, cam.position.y, cam.position.z = 0.25, -0.5, 1.0
            r.set_camera(cam)
            o.set_camera(cam)
        r.generate_frame()
        o.generate_frame()
        assert np.array_equal(r.live_counts(), o.live_counts()), (seed, tick)
    assert np.array_equal(r.accumulator(), o.accumulator())
    assert np.array_equal(r.pixels(), o.pixels())
    assert np.array_equal(r.float_accumulator(), o.float_sum(), equal_nan=True)
    assert r.total_ray_bounces() == o.total_ray_bounces()
    r.close()
    o.close()
