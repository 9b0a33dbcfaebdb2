"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on a CPU-only box.

The oracle (oracle/) is test infrastructure: it is imported here and nowhere in the product package.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _ensure_built():
    # libraries are built in-tree by __graft_entry__.build(); build them here only if a fresh checkout lacks them
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "cudapathtracer_amd", "libmipt.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cudapathtracer_amd", "csrc")])


_ensure_built()

import cudapathtracer_amd as pt  # noqa: E402
from oracle import orc  # noqa: E402


@pytest.fixture(scope="session")
def ptlib():
    return pt


@pytest.fixture(scope="session")
def oracle():
    return orc


@pytest.fixture(scope="session")
def tracer():
    """One tracer (device 0) for all GPU tests of the session."""
    t = pt.Tracer(0)
    yield t
    t.close()


def random_rays(n, seed, lo=-3.0, hi=3.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def triangle_soup(n, seed, extent=4.0, size=0.4):
    """n random small triangles as one mesh (vertices, indices)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    p = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32).reshape(-1, 3)
    v = np.zeros(3 * n, pt.VERTEX_DTYPE)
    v["position"] = p
    nrm = np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    v["normal"] = np.repeat(nrm, 3, axis=0).astype(np.float32)
    return v, np.arange(3 * n, dtype=np.uint32)


def soup_scene(n, seed, **mat):
    v, i = triangle_soup(n, seed)
    sky = pt.SkyLight()
    sky.color[:] = (0.98, 0.97, 0.99)
    sky.dir[:] = (-0.57735026, -0.57735026, -0.57735026)
    sky.intensity = 5.0
    return pt.SceneData([(v, i, pt.make_material(**mat))], sky)


def hits_equal(a, b):
    """bit-exact comparison of two hit arrays; returns number of differing records"""
    bad = np.zeros(len(a), bool)
    for f in ("t", "u", "v"):
        bad |= a[f].view(np.uint32) != b[f].view(np.uint32)
    for f in ("tri", "mesh", "front"):
        bad |= a[f] != b[f]
    return int(bad.sum())
