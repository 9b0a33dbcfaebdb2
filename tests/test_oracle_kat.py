"""Pins the oracle: RNG known answers (BASELINE.md section 3 / SURVEY 8c), analytic ray-triangle / ray-box answers,
and the portable transcendentals against glibc.  The reference ships no golden vectors for this path ("parity
unpinned by the reference"): these known answers are all that exists."""
import numpy as np
import pytest

from conftest import random_rays

KAT = [  # (v0, v1, TEA16 state, first three LCG floats) -- transcription of CUDATracerProg.cu:40-72
    (0, 0, 0x741c187d, (0.824214458, 0.00487846136, 0.754297853)),
    (1, 0, 0x8da6b311, (0.905536413, 0.431570709, 0.667385936)),
    (0, 1, 0x70d3aef1, (0.421319723, 0.6455428, 0.56328094)),
    (32896, 0, 0x8a1c9bab, (0.609330058, 0.548356354, 0.293936729)),
    (1037760, 7, 0xc908dec6, (0.98107034, 0.540556431, 0.126406133)),
]


@pytest.mark.parametrize("v0,v1,state,floats", KAT)
def test_tea_lcg_known_answers(oracle, v0, v1, state, floats):
    assert oracle.tea16(v0, v1) == state
    got = oracle.lcg_floats(state, 3)
    assert np.array_equal(got, np.array(floats, np.float32))
    assert ((got >= 0) & (got < 1)).all()


def test_lcg_low_24_bits_only(oracle):
    # NextFloat = (state & 0xFFFFFF) / 2^24 : every value is a multiple of 2^-24 in [0,1)
    f = oracle.lcg_floats(12345, 4096).astype(np.float64) * 2 ** 24
    assert np.array_equal(f, np.round(f)) and f.max() < 2 ** 24


def test_node_seed_distinct(oracle):
    seeds = {oracle.lib.orc_node_seed(0x741c187d, n) for n in range(1, 4096)}
    assert len(seeds) == 4095


def ulp_diff(a, b):
    a = a.astype(np.float32).view(np.int32).astype(np.int64)
    b = b.astype(np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


def test_portable_math_close_to_libm(oracle):
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2 * np.pi, 200000).astype(np.float32)
    for fn in (0, 1):   # sin, cos on [0, 2pi]: absolute error (values cross zero)
        d = oracle.math_eval(fn, oracle.MATH_DET, x)
        m = oracle.math_eval(fn, oracle.MATH_LIBM, x)
        assert np.abs(d - m).max() < 2.5e-7
    y, xx = rng.uniform(-2, 2, 200000).astype(np.float32), rng.uniform(-2, 2, 200000).astype(np.float32)
    assert ulp_diff(oracle.math_eval(2, oracle.MATH_DET, y, xx), oracle.math_eval(2, oracle.MATH_LIBM, y, xx)).max() <= 4
    a = rng.uniform(-1, 1, 200000).astype(np.float32)
    assert np.abs(oracle.math_eval(3, oracle.MATH_DET, a) - oracle.math_eval(3, oracle.MATH_LIBM, a)).max() < 5e-7
    g = rng.uniform(0, 1, 200000).astype(np.float32)
    assert ulp_diff(oracle.math_eval(4, oracle.MATH_DET, g), oracle.math_eval(4, oracle.MATH_LIBM, g)).max() <= 8
    p = rng.uniform(0, 2, 200000).astype(np.float32)
    assert ulp_diff(oracle.math_eval(5, oracle.MATH_DET, p), oracle.math_eval(5, oracle.MATH_LIBM, p)).max() <= 4
    q = rng.uniform(0, 1, 200000).astype(np.float32)
    assert ulp_diff(oracle.math_eval(6, oracle.MATH_DET, q), oracle.math_eval(6, oracle.MATH_LIBM, q)).max() <= 8   # 4 roundings amplified by the squarings


def test_portable_math_special_values(oracle):
    z = np.array([0.0], np.float32)
    assert oracle.math_eval(4, oracle.MATH_DET, z)[0] == 0.0                       # 0^(1/2.2)
    assert oracle.math_eval(3, oracle.MATH_DET, np.array([1.0], np.float32))[0] == 0.0    # acos(1)
    assert abs(oracle.math_eval(3, oracle.MATH_DET, np.array([-1.0], np.float32))[0] - np.pi) < 1e-6
    assert oracle.math_eval(2, oracle.MATH_DET, z, z)[0] == 0.0                    # atan2(0,0)
    assert abs(oracle.math_eval(2, oracle.MATH_DET, np.array([1.0], np.float32), z)[0] - np.pi / 2) < 1e-7
    assert abs(oracle.math_eval(2, oracle.MATH_DET, z, np.array([-1.0], np.float32))[0] - np.pi) < 1e-6


def one_triangle_scene(ptlib):
    v = np.zeros(3, ptlib.VERTEX_DTYPE)
    v["position"] = [(0, 0, 0), (0, 0, 1), (1, 0, 1)]        # the first triangle of MakeQuad: normal +y
    v["normal"] = [(0, 1, 0)] * 3
    sky = ptlib.SkyLight()
    return ptlib.SceneData([(v, np.array([0, 1, 2], np.uint32), ptlib.make_material())], sky)


def test_ray_triangle_known_answers(ptlib, oracle):
    sc = oracle.Scene(one_triangle_scene(ptlib))
    o = np.array([(0.25, 1, 0.5), (0.25, -1, 0.5), (0.25, 1, 0.5), (2, 1, 2), (0.25, 1, 0.5), (0.25, 0, 0.5)], np.float32)
    d = np.array([(0, -1, 0), (0, 1, 0), (0, 1, 0), (0, -1, 0), (1, 0, 0), (0, -1, 0)], np.float32)
    for mode in (oracle.ISECT_REF, oracle.ISECT_BRUTE):
        h, _ = sc.trace_rays(mode, o, d, nthreads=1)
        assert h["tri"].tolist() == [0, 0, -1, -1, -1, -1]     # behind / outside / parallel / t == 0 all miss
        assert h["t"][0] == 1.0 and h["t"][1] == 1.0
        assert h["front"][0] == 1 and h["front"][1] == 0        # front = det > 0 = ray against the +y normal
        # barycentrics: P = (0.25, 0, 0.5) = v0 + u*e1 + v*e2 with e1=(0,0,1), e2=(1,0,1): v = 0.25, u = 0.25
        assert h["u"][0] == 0.25 and h["v"][0] == 0.25


def test_ref_and_brute_agree_on_reference_scenes(ptlib, oracle):
    for which in (0, 1):
        sc = oracle.Scene(ptlib.builtin_scene(which))
        o, d = random_rays(30000, 11 + which)
        a, _ = sc.trace_rays(oracle.ISECT_REF, o, d)
        b, _ = sc.trace_rays(oracle.ISECT_BRUTE, o, d)
        diff = np.nonzero((a["tri"] != b["tri"]) | (a["t"] != b["t"]))[0]
        # the median BVH may only differ from brute force on exact-t ties (coplanar cube bottoms / floor) or on
        # knife-edge box misses; both are rare
        assert len(diff) <= 30, len(diff)
        assert (ulp_diff(a["t"][diff], b["t"][diff]) <= 4).sum() >= len(diff) - 3
