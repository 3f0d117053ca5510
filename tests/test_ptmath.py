"""Shared deterministic math (csrc/ptmath.h) pinned against libm in float64: the tracer's parity
argument needs these functions to be (a) the same bits on host and device — by construction —
and (b) as accurate as the CUDA libm the reference used (a few ulp)."""
import ctypes as C
import math

import numpy as np
import pytest

import ptss

_f32p = C.POINTER(C.c_float)
OPS = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "log": 4, "exp": 5, "pow": 6, "sqrt": 7}


def ev(op, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = np.ascontiguousarray(y, np.float32) if y is not None else None
    rc = ptss.host_lib().ptss_probe_math(OPS[op], x.ctypes.data_as(_f32p),
                                         yy.ctypes.data_as(_f32p) if yy is not None else None,
                                         out.ctypes.data_as(_f32p), x.size)
    assert rc == 0
    return out


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    return np.abs(got.astype(np.float64) - ref64) / np.spacing(np.abs(ref32)).astype(np.float64)


RNG = np.random.default_rng(20261004)


def test_sincos_on_sampler_range():
    x = RNG.uniform(0, 2 * math.pi, 400_000).astype(np.float32)  # theta = u * 2 * pi (CudaTracer.cu:536)
    assert ulp_err(ev("sin", x), np.sin(x.astype(np.float64))).max() <= 2.0
    assert ulp_err(ev("cos", x), np.cos(x.astype(np.float64))).max() <= 2.0
    x = RNG.uniform(-60, 60, 200_000).astype(np.float32)
    assert np.abs(ev("sin", x) - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(ev("cos", x) - np.cos(x.astype(np.float64))).max() < 2e-7


def test_sincos_exact_points_and_pythagoras():
    assert ev("sin", [0.0])[0] == 0.0 and ev("cos", [0.0])[0] == 1.0
    x = RNG.uniform(0, 2 * math.pi, 100_000).astype(np.float32)
    s, c = ev("sin", x).astype(np.float64), ev("cos", x).astype(np.float64)
    assert np.abs(s * s + c * c - 1).max() < 3e-7


def test_atan():
    x = np.concatenate([RNG.uniform(0, 30, 200_000), 10 ** RNG.uniform(-6, 6, 200_000)]).astype(np.float32)
    assert ulp_err(ev("atan", x), np.arctan(x.astype(np.float64))).max() <= 3.0
    assert np.array_equal(ev("atan", -x), -ev("atan", x))
    r = ev("atan", [np.inf, -np.inf, 0.0])
    assert r[0] == np.float32(math.pi / 2) and r[1] == -np.float32(math.pi / 2) and r[2] == 0


def test_log():
    x = np.concatenate([RNG.uniform(0, 1, 300_000), 10 ** RNG.uniform(-44, 4, 100_000)]).astype(np.float32)
    x = x[x > 0]
    assert ulp_err(ev("log", x), np.log(x.astype(np.float64))).max() <= 1.5
    r = ev("log", [0.0, -1.0, np.inf, 1.0])
    assert r[0] == -np.inf and np.isnan(r[1]) and r[2] == np.inf and r[3] == 0.0


def test_exp():
    x = RNG.uniform(-87, 10, 400_000).astype(np.float32)
    assert ulp_err(ev("exp", x), np.exp(x.astype(np.float64))).max() <= 1.5
    # Beer-Lambert uses exp(-d*a) <= 1: the implementation must never exceed 1 for x <= 0
    x = -(10 ** RNG.uniform(-10, 2, 200_000)).astype(np.float32)
    assert (ev("exp", x) <= 1.0).all()
    r = ev("exp", [0.0, -200.0, 100.0, -np.inf])
    assert r[0] == 1.0 and r[1] == 0.0 and r[2] == np.inf and r[3] == 0.0


def test_pow_gamma_and_phong():
    x = RNG.uniform(0, 1, 400_000).astype(np.float32)
    g = np.float32(1 / 2.2)
    r = ev("pow", x, np.full_like(x, g))
    ref = x.astype(np.float64) ** np.float64(g)
    assert (np.abs(r - ref) / np.maximum(ref, 1e-30)).max() < 1e-6
    assert r.max() <= 1.0
    for e in (250.0, 300.0):  # Scene.cpp:101-105 Phong exponents: y = pow(s, 1/(e+1)) must stay <= 1
        y = np.float32(1) / np.float32(e + 1)
        r = ev("pow", x, np.full_like(x, y))
        assert r.max() <= 1.0
        assert ulp_err(r, x.astype(np.float64) ** np.float64(y)).max() <= 2.0
    assert ev("pow", [0.0, 1.0, 0.5], [g, g, 0.0]).tolist() == [0.0, 1.0, 1.0]


def test_nan_policy():
    for op in ("sin", "cos", "atan", "log", "exp"):
        assert np.isnan(ev(op, [np.nan]))[0], op
    assert np.isnan(ev("sin", [np.inf, 1e9])).all()  # outside the reduction range -> NaN, never garbage
    assert np.isnan(ev("pow", [-1.0, np.nan], [0.5, 0.5])).all()


def test_tracer_constants():
    # s = -2 * tan(fov / 2) with fov = pi/2 (CudaTracer.cu:334): within 1 ulp of 1
    t = ev("tan", [np.float32(math.pi) / np.float32(2) * np.float32(0.5)])[0]
    assert abs(float(t) - 1.0) <= 1.2e-7
    assert ev("sqrt", [2.0])[0] == np.float32(math.sqrt(2.0))


def test_tone_map_threshold_table_describes_the_literal_function():
    """csrc/ptquant.h: T[k] is the smallest float whose 8-bit sample is >= k. Checked here on the host: the table is
    strictly increasing, each threshold and its predecessor float sit on the two sides of a step, and resolving random
    radiances through the table (searchsorted) reproduces the literal clamp / pow / scale function. The device form
    (hardware guess + two compares) is proven against the literal one for all 2^32 patterns in tests/test_gpu_math.py."""
    import ctypes as C
    L = ptss.host_lib()
    T = np.zeros(257, np.float32)
    assert L.ptss_probe_quant_table(T.ctypes.data_as(_f32p)) == 0
    assert T[0] == -np.inf and np.isnan(T[256]) and np.all(np.diff(T[1:256]) > 0) and 0 < T[1] and T[255] < 1

    def literal(x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(x.size, np.uint32)
        assert L.ptss_probe_quantize(x.ctypes.data_as(_f32p), out.ctypes.data_as(C.POINTER(C.c_uint)), x.size) == 0
        return out

    at = T[1:256]
    below = np.nextafter(at, np.float32(-1), dtype=np.float32)
    assert np.array_equal(literal(at), np.arange(1, 256)) and np.array_equal(literal(below), np.arange(0, 255))
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(200000, np.float32), rng.random(50000, np.float32) ** 8, np.float32([0, 1, -1, 2, np.inf, -np.inf, np.nan, 1e-30, 0.5])])
    want = literal(x)
    via_table = np.where(np.isnan(x), 0, np.searchsorted(T[1:256], x, side="right")).astype(np.uint32)
    assert np.array_equal(via_table, want)
    assert want[-3] == 0 and want[200000 + 50000 + 1] == 255 and want[200000 + 50000 + 6] == 0  # 1e-30, 1.0, NaN
