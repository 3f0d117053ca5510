"""XORWOW: the product's generator (csrc/xorwow.h, probed through libptss_host), the oracle's own
statement of it, Marsaglia's published recurrence, and rocRAND's independent precomputed 2^67
jump matrices must all agree."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
import ptss

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


def product_init(seed, subseq):
    out = np.empty(6, np.uint32)
    assert ptss.host_lib().ptss_probe_rng_init(seed, subseq, out.ctypes.data_as(_u32p)) == 0
    return out


def product_draw(state, n):
    st = state.copy()
    raw = np.empty(n, np.uint32)
    uni = np.empty(n, np.float32)
    assert ptss.host_lib().ptss_probe_rng_draw(st.ctypes.data_as(_u32p), raw.ctypes.data_as(_u32p),
                                               uni.ctypes.data_as(_f32p), n) == 0
    return st, raw, uni


def marsaglia(v, d, n):
    """xorwow as published (Marsaglia 2003, 'Xorshift RNGs', section 3.1), pure Python."""
    v = [int(x) for x in v]
    d = int(d)
    out = []
    M = 0xFFFFFFFF
    for _ in range(n):
        t = v[0] ^ (v[0] >> 2)
        v = v[1:] + [(v[4] ^ ((v[4] << 4) & M)) ^ (t ^ ((t << 1) & M))]
        d = (d + 362437) & M
        out.append((v[4] + d) & M)
    return out


def test_step_matches_published_recurrence():
    st = product_init(0x5EED, 0)
    _, raw, _ = product_draw(st, 64)
    assert raw.tolist() == marsaglia(st[:5], st[5], 64)


def test_seed_zero_state_is_curand_scramble():
    # curand_init(0, 0, 0): s0 = 0xaad26b49, s1 = 0xf7dcefdd
    t0 = (1099087573 * 0xaad26b49) & 0xFFFFFFFF
    t1 = (2591861531 * 0xf7dcefdd) & 0xFFFFFFFF
    want = [(123456789 + t0) & 0xFFFFFFFF, 362436069 ^ t0, (521288629 + t1) & 0xFFFFFFFF, 88675123 ^ t1,
            (5783321 + t0) & 0xFFFFFFFF, (6615241 + t1 + t0) & 0xFFFFFFFF]
    assert product_init(0, 0).tolist() == want


@pytest.mark.parametrize("seed,subseq", [(0x5EED, 0), (0x5EED, 1), (0x5EED, 65535), (1234567890123, 2073599),
                                         (0, 8294399), (7, 0xFFFFFFFF)])
def test_oracle_and_product_agree(seed, subseq):
    ost, oraw, ouni = oracle.probe_rng(seed, subseq, 64)
    pst = product_init(seed, subseq)
    assert ost.tolist() == pst.tolist()
    _, praw, puni = product_draw(pst, 64)
    assert np.array_equal(oraw, praw)
    assert np.array_equal(ouni, puni)


def test_uniform_is_half_open_at_zero():
    st = product_init(0x5EED, 3)
    _, raw, uni = product_draw(st, 4096)
    assert (uni > 0).all() and (uni <= 1).all()
    # pinned mapping: x * 2^-32 + 2^-33 in float32 (two roundings)
    ref = raw.astype(np.float32) * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)
    assert np.array_equal(uni, ref)


def test_subsequences_compose():
    # skipping a+b subsequences == skipping a then b: check through the jump table itself
    t = np.zeros(32 * 800, np.uint32)
    assert ptss.host_lib().ptss_probe_rng_jump_table(t.ctypes.data_as(_u32p), t.size) == 0
    t = t.reshape(32, 160, 5)

    def apply(m, v):
        r = np.zeros(5, np.uint32)
        for i in range(160):
            if (int(v[i >> 5]) >> (i & 31)) & 1:
                r ^= m[i]
        return r

    v = product_init(42, 0)[:5]
    v3 = apply(t[1], apply(t[0], v))  # 1 + 2 = 3
    assert product_init(42, 3)[:5].tolist() == v3.tolist()
    # squaring: table[k+1] == table[k] applied twice, on a probe vector
    for k in (0, 5, 17, 30):
        assert apply(t[k + 1], v).tolist() == apply(t[k], apply(t[k], v)).tolist()


ROCRAND_TABLE = "/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h"


@pytest.mark.skipif(not os.path.exists(ROCRAND_TABLE), reason="rocRAND headers not installed")
def test_jump_table_matches_rocrand_precomputed():
    """rocRAND ships A^(4^k * 2^67) for k = 0..31 as literal tables (h_xorwow_sequence_jump_matrices);
    ours are A^(2^(67+k)) computed by 67+k squarings: ours[2k] must equal theirs[k]."""
    text = open(ROCRAND_TABLE).read()
    m = re.search(r"h_xorwow_sequence_jump_matrices\[XORWOW_JUMP_MATRICES\]\[XORWOW_SIZE\]\s*=\s*\{(.*?)\};", text, re.S)
    assert m, "table not found in rocRAND header"
    nums = np.array([int(x) for x in re.findall(r"\d+", re.sub(r"//.*", "", m.group(1)))], dtype=np.uint64)
    assert nums.size == 32 * 800
    theirs = nums.astype(np.uint32).reshape(32, 800)
    t = np.zeros(32 * 800, np.uint32)
    assert ptss.host_lib().ptss_probe_rng_jump_table(t.ctypes.data_as(_u32p), t.size) == 0
    ours = t.reshape(32, 800)
    for k in range(16):
        assert np.array_equal(ours[2 * k], theirs[k]), f"A^(2^{67 + 2 * k}) differs from rocRAND"
