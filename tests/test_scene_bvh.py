"""Callers' scene construction (product C++ vs the oracle's independent restatement) and the product's BVH builders,
checked structurally and functionally with the oracle.  CPU only: host-only scenes need no GPU."""
import os

import numpy as np
import pytest

from conftest import random_rays, soup_scene
from test_oracle_kat import ulp_diff


@pytest.mark.parametrize("which,ntris,nmesh", [(0, 36, 8), (1, 12506, 5)])
def test_builtin_scenes_identical_to_oracle_restatement(ptlib, oracle, which, ntris, nmesh):
    a = ptlib.builtin_scene(which)
    b = oracle.builtin_meshes(which)
    assert len(a.meshes) == len(b) == nmesh and a.triangle_count == ntris
    for (v, i, m), (v2, i2, m2) in zip(a.meshes, b):
        assert v.tobytes() == v2.tobytes() and i.tobytes() == i2.tobytes() and bytes(m) == bytes(m2)
    assert bytes(a.sky) == bytes(oracle.builtin_sky(which))


def test_sphere_and_cube_counts(ptlib):
    import ctypes as C
    for kind, nv, ni in ((0, 4, 6), (1, 24, 36), (2, 3122, 6240 * 3)):
        icnt = C.c_uint32(0)
        assert ptlib.lib.mipt_make_mesh(kind, None, None, 0, None, 0, C.byref(icnt)) == nv
        assert icnt.value == ni


def test_cornell_layout(ptlib):
    # light quad 1e-5 below the ceiling, red wall at z=-2 (viewer's left), camera looks down +x
    d = ptlib.builtin_scene(0)
    light = d.meshes[5][0]["position"]
    assert np.allclose(light[:, 1], 1.99999, atol=1e-6) and np.abs(light[:, [0, 2]]).max() <= 0.5 + 1e-6
    assert d.meshes[5][2].emissiveStrength == 10.0
    assert tuple(d.meshes[3][2].color) == (1.0, 0.0, 0.0) and tuple(d.meshes[4][2].color) == (0.0, 1.0, 0.0)
    allp = np.concatenate([m[0]["position"] for m in d.meshes])
    assert np.allclose(allp.min(0), -2, atol=1e-5) and np.allclose(allp.max(0), 2, atol=1e-5)
    assert d.sky.intensity == 0.0


def test_camera_matches_oracle(ptlib, oracle):
    for kw in (ptlib.CORNELL_CAMERA, ptlib.DEFAULT_CAMERA, dict(pos=(1, 2, 3), azimuth=0.7, zenith=1.1, fov=0.9)):
        a, b = ptlib.make_camera(**kw), oracle.make_camera(**kw)
        assert bytes(a) == bytes(b)
    c = ptlib.make_camera(**ptlib.CORNELL_CAMERA)
    assert abs(c.front[0] - 1) < 1e-6 and abs(c.right[2] - 1) < 1e-6 and abs(c.up[1] - 1) < 1e-6


def decode(ref):
    return bool(ref & 0x80000000), (ref >> 27) & 0xF, ref & 0x07FFFFFF


@pytest.mark.parametrize("builder", [0, 1])
@pytest.mark.parametrize("scene", ["cornell", "default", "soup"])
@pytest.mark.parametrize("width", [2, 4], ids=["bvh2", "q4"])
def test_flat_bvh_valid_and_equal_to_brute_force(ptlib, oracle, builder, scene, width):
    data = {"cornell": lambda: ptlib.builtin_scene(0), "default": lambda: ptlib.builtin_scene(1),
            "soup": lambda: soup_scene(10000, 5)}[scene]()
    sc = ptlib.Scene(None, data, builder=builder, node_width=width)
    nodes, refs = sc.bvh()
    assert sc.bvh_width == width and len(refs) >= data.triangle_count
    # auto layout: BVH2 when nodes + references fit the kernels' 4 KiB LDS window (the Cornell box), else 4-wide
    assert ptlib.Scene(None, data, builder=builder).bvh_width == (2 if scene == "cornell" else 4)
    o = oracle.Scene(data)
    o.set_flat_bvh(nodes, refs, width)
    assert o.check_flat_bvh() == (0, "")
    # leaf sizes respect the default maximum of 4; children boxes are well formed
    kids = nodes[:, 12:12 + width].view(np.uint32)
    leaf = (kids & 0x80000000) != 0
    assert (((kids >> 27) & 0xF)[leaf] <= 4).all()
    if width == 2:
        assert (nodes[:, 0:3] <= nodes[:, 3:6]).all() and (nodes[:, 6:9] <= nodes[:, 9:12]).all()
    else:
        q = nodes[:, 4:10].view(np.uint8).reshape(-1, 6, 4)                  # [node][qlo xyz, qhi xyz][child]
        used = (kids != 0x80000000)
        assert ((q[:, 0:3, :] <= q[:, 3:6, :]) | ~used[:, None, :]).all()
        assert len(nodes) < len(ptlib.Scene(None, data, builder=builder, node_width=2).bvh()[0])
    og, dg = random_rays(20000, 21)
    a, st = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
    diff = np.nonzero((a["tri"] != b["tri"]) | (a["t"] != b["t"]))[0]
    assert len(diff) <= 40, len(diff)                     # coplanar ties within an ulp / knife-edge box misses only
    assert (ulp_diff(a["t"][diff], b["t"][diff]) <= 4).sum() >= len(diff) - 3
    assert st.node_visits > 0 and st.tri_tests > 0


def test_sbvh_spatial_splits_duplicate_references(ptlib, oracle):
    # two huge ground triangles under 2000 small ones: an object split must put the ground in one child and blow
    # its box up; the spatial split chops the ground instead and duplicates its references
    rng = np.random.default_rng(2)
    n = 2002
    c = rng.uniform(-4, 4, (n - 2, 1, 3)) * (1, 0, 1) + (0, 0.25, 0)
    small = c + rng.uniform(-0.1, 0.1, (n - 2, 3, 3))
    ground = np.array([[(-4, 0, -4), (-4, 0, 4), (4, 0, 4)], [(-4, 0, -4), (4, 0, 4), (4, 0, -4)]], np.float64)
    v = np.zeros(3 * n, ptlib.VERTEX_DTYPE)
    v["position"] = np.concatenate([ground, small]).astype(np.float32).reshape(-1, 3)
    v["normal"] = (0, 1, 0)
    data = ptlib.SceneData([(v, np.arange(3 * n, dtype=np.uint32), ptlib.make_material())], ptlib.SkyLight())
    nodes, refs = ptlib.Scene(None, data, node_width=2).bvh()
    assert len(refs) > n
    o = oracle.Scene(data)
    o.set_flat_bvh(nodes, refs)
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(20000, 4, -4, 4)
    og[:, 1] = np.abs(og[:, 1]) * 0.2
    f, sf = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
    assert ((f["tri"] != b["tri"]) | (f["t"] != b["t"])).sum() <= 10
    # and it pays: fewer triangle tests than the median tree on the same rays
    n2, r2 = ptlib.Scene(None, data, builder=1, node_width=2).bvh()
    o.set_flat_bvh(n2, r2)
    _, sm = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    assert sf.tri_tests < sm.tri_tests


@pytest.mark.parametrize("ntri", [0, 1, 2, 5])
def test_tiny_scenes(ptlib, oracle, ntri):
    data = soup_scene(max(ntri, 1), 9)
    if ntri == 0:
        data = ptlib.SceneData([], data.sky)
    for builder, width in ((0, 2), (1, 2), (0, 4), (1, 4), (0, 0)):
        nodes, refs, w = ptlib.Scene(None, data, builder=builder, node_width=width).flat()
        assert len(nodes) >= 1 and len(refs) == ntri and w == (width or 2)
        o = oracle.Scene(data)
        o.set_flat_bvh(nodes, refs, w)
        assert o.check_flat_bvh()[0] == 0
        og, dg = random_rays(2000, 1, -5, 5)
        f, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg)
        b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
        assert np.array_equal(f["tri"], b["tri"]) and np.array_equal(f["t"], b["t"])


def test_degenerate_and_duplicate_triangles(ptlib, oracle):
    v = np.zeros(3 * 40, ptlib.VERTEX_DTYPE)
    pos = np.zeros((40, 3, 3), np.float32)
    pos[:20] = [(0, 0, 0), (0, 0, 1), (1, 0, 1)]            # 20 identical triangles
    pos[20:30] = [(1, 1, 1), (1, 1, 1), (1, 1, 1)]          # zero-area
    pos[30:] = [(0, 2, 0), (1, 2, 0), (2, 2, 0)]            # collinear
    v["position"] = pos.reshape(-1, 3)
    v["normal"] = (0, 1, 0)
    data = ptlib.SceneData([(v, np.arange(120, dtype=np.uint32), ptlib.make_material())], ptlib.SkyLight())
    for builder, width in ((0, 2), (1, 2), (0, 4), (1, 4)):
        nodes, refs, w = ptlib.Scene(None, data, builder=builder, node_width=width).flat()
        o = oracle.Scene(data)
        o.set_flat_bvh(nodes, refs, w)
        assert o.check_flat_bvh()[0] == 0
        h, _ = o.trace_rays(oracle.ISECT_FLAT, np.array([(0.25, 1, 0.5)], np.float32), np.array([(0, -1, 0)], np.float32))
        assert h["tri"][0] == 0 and h["t"][0] == 1.0        # lowest id wins the 20-way tie


def test_grid_mesh_and_synthetic_sky(ptlib):
    v, i = ptlib.grid_mesh(32, 8.0, 1.0, 1)
    assert len(v) == 33 * 33 and len(i) == 32 * 32 * 6 and i.max() == len(v) - 1
    p = v["position"]
    assert p[:, 0].min() == -8 and p[:, 0].max() == 8 and 0 <= p[:, 1].min() and p[:, 1].max() <= 1
    # triangles face +y
    t = p[i.reshape(-1, 3)]
    assert (np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])[:, 1] > 0).all()
    sky = ptlib.synthetic_sky(256, 128, 0x5EED)
    assert sky.shape == (128, 256, 3) and sky.min() >= 0 and sky.max() == 5.0e4
    assert np.array_equal(sky, ptlib.synthetic_sky(256, 128, 0x5EED))
    assert sky[:64].mean() > sky[64:].mean()               # bright sky above, dark ground below


def test_parallel_build_is_thread_count_independent(ptlib, oracle, monkeypatch):
    # 131 k triangles: above the builder's task threshold, so right subtrees are built by child builders and spliced
    v, i = ptlib.grid_mesh(256, 8.0, 1.0, 1)
    data = ptlib.SceneData([(v, i, ptlib.make_material())], ptlib.SkyLight())
    out = []
    for threads in ("1", "2", "7"):
        monkeypatch.setenv("MIPT_BUILD_THREADS", threads)
        nodes, refs, width = ptlib.Scene(None, data).flat()
        assert width == 4
        out.append((nodes.copy(), refs.copy()))
    for nodes, refs in out[1:]:
        assert np.array_equal(nodes.view(np.uint32), out[0][0].view(np.uint32)) and np.array_equal(refs, out[0][1])
    nodes, refs = out[0]
    o = oracle.Scene(data)
    o.set_flat_bvh(nodes, refs, 4)
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(20000, 33, -6, 6)
    og[:, 1] = 2 + np.abs(og[:, 1]) * 0.3
    dg[:, 1] = -np.abs(dg[:, 1])
    f, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
    assert (f["tri"] >= 0).sum() > 5000
    assert ((f["tri"] != b["tri"]) | (f["t"] != b["t"])).sum() <= 10


def test_bvh_cache_roundtrip_and_rejection(ptlib, oracle, tmp_path):
    data = ptlib.builtin_scene(1)
    path = str(tmp_path / "default.bvh")
    a = ptlib.Scene(None, data, bvh_cache=path)
    assert not a.bvh_was_cached and os.path.getsize(path) > 32
    b = ptlib.Scene(None, data, bvh_cache=path)
    assert b.bvh_was_cached
    na, ra = a.bvh(); nb, rb = b.bvh()
    assert np.array_equal(na.view(np.uint32), nb.view(np.uint32)) and np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
    # other build parameters or other triangles: the file does not match, the BVH is rebuilt (and the file replaced)
    c = ptlib.Scene(None, data, bvh_cache=path, max_leaf=2)
    assert not c.bvh_was_cached
    assert ptlib.Scene(None, data, bvh_cache=path, max_leaf=2).bvh_was_cached
    assert not ptlib.Scene(None, ptlib.builtin_scene(0), bvh_cache=path).bvh_was_cached
    # corrupt structure under a matching key: rejected by the validator, rebuilt
    d = ptlib.Scene(None, data, bvh_cache=path)
    assert not d.bvh_was_cached
    raw = bytearray(open(path, "rb").read())
    hdr = 32
    for bad in (np.uint32(0x0FFFFFFF), np.uint32(0), np.uint32(0x80000000 | (15 << 27) | 0x07FFFFF0)):
        mod = bytearray(raw)
        mod[hdr + 12 * 4: hdr + 13 * 4] = bad.tobytes()        # root's first child: out of range / cycle / leaf past the end
        open(path, "wb").write(mod)
        e = ptlib.Scene(None, data, bvh_cache=path)
        assert not e.bvh_was_cached
        ne, _ = e.bvh()
        assert np.array_equal(ne.view(np.uint32), na.view(np.uint32))
    open(path, "wb").write(raw[:100])                            # truncated
    assert not ptlib.Scene(None, data, bvh_cache=path).bvh_was_cached


def _moved(ptlib, data, amp, seed=0):
    """same topology, every vertex displaced by a smooth field + noise; normals left alone"""
    rng = np.random.default_rng(seed)
    meshes = []
    for v, i, m in data.meshes:
        v2 = v.copy()
        p = v2["position"].astype(np.float64)
        p = p + amp * np.stack([np.sin(1.3 * p[:, 1] + 0.2), np.cos(0.9 * p[:, 2]), np.sin(1.1 * p[:, 0])], 1) + rng.uniform(-amp, amp, p.shape) * 0.2
        v2["position"] = p.astype(np.float32)
        meshes.append((v2, i, m))
    return ptlib.SceneData(meshes, data.sky, data.env)


@pytest.mark.parametrize("builder", [0, 1])
@pytest.mark.parametrize("width", [2, 4], ids=["bvh2", "q4"])
@pytest.mark.parametrize("scene", ["default", "soup"])
def test_refit_keeps_the_tree_and_encloses_the_moved_triangles(ptlib, oracle, builder, width, scene):
    """mipt_scene_refit (include/mipt.h): the child references stay, the boxes follow the vertices, and traversal of the
    re-fitted BVH finds what brute force finds on the moved geometry."""
    data = ptlib.builtin_scene(1) if scene == "default" else soup_scene(6000, 8)
    sc = ptlib.Scene(None, data, builder=builder, node_width=width)
    n0, r0 = sc.bvh()
    moved = _moved(ptlib, data, 0.35)
    sc.refit(moved)
    n1, r1 = sc.bvh()
    assert n1.shape == n0.shape and r1.shape == r0.shape and sc.bvh_width == width
    assert np.array_equal(n1[:, 12:16].view(np.uint32), n0[:, 12:16].view(np.uint32))            # same tree
    assert np.array_equal(r1[:, [3, 7]].view(np.uint32), r0[:, [3, 7]].view(np.uint32))          # same reference order
    assert not np.array_equal(r1[:, 0:3], r0[:, 0:3])
    o = oracle.Scene(moved)
    o.set_flat_bvh(n1, r1, width)
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(20000, 33)
    a, st = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
    diff = np.nonzero((a["tri"] != b["tri"]) | (a["t"] != b["t"]))[0]
    assert len(diff) <= 40 and (ulp_diff(a["t"][diff], b["t"][diff]) <= 4).sum() >= len(diff) - 3
    # a tree built for the moved geometry is cheaper to traverse, but the re-fitted one is not absurdly worse
    fresh = ptlib.Scene(None, moved, builder=builder, node_width=width)
    o.set_flat_bvh(*fresh.flat())
    _, sf = o.trace_rays(oracle.ISECT_FLAT, og, dg)
    assert st.node_visits < 6 * sf.node_visits
    # moving back: the median builder's boxes (no clipped spatial-split leaves) are reproduced exactly in the BVH2 layout
    sc.refit(data)
    n2, r2 = sc.bvh()
    assert np.array_equal(r2, r0)
    if builder == 1 and width == 2:
        assert np.array_equal(n2, n0)
    o2 = oracle.Scene(data)
    o2.set_flat_bvh(n2, r2, width)
    assert o2.check_flat_bvh() == (0, "")


def test_refit_rejects_another_topology_and_handles_tiny_scenes(ptlib, oracle):
    data = soup_scene(50, 1)
    sc = ptlib.Scene(None, data)
    before = sc.bvh()
    with pytest.raises(ptlib.MiptError):
        sc.refit(soup_scene(51, 1))
    v, i, m = data.meshes[0]
    with pytest.raises(ptlib.MiptError):
        sc.refit(ptlib.SceneData([(v, i, m), (v, i, m)], data.sky))
    bad = i.copy(); bad[5] = len(v)
    with pytest.raises(ptlib.MiptError):
        sc.refit(ptlib.SceneData([(v, bad, m)], data.sky))
    # failed refits left the scene alone (compared as bit patterns: the plane words of empty child slots read as NaN)
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(before, sc.bvh()))
    empty = ptlib.Scene(None, ptlib.SceneData([], data.sky))
    empty.refit(ptlib.SceneData([], data.sky))
    for width in (2, 4):
        one = soup_scene(1, 2)
        s1 = ptlib.Scene(None, one, node_width=width)
        mv = _moved(ptlib, one, 2.0)
        s1.refit(mv)
        o = oracle.Scene(mv)
        o.set_flat_bvh(*s1.flat())
        assert o.check_flat_bvh()[0] == 0
        og, dg = random_rays(3000, 1, -5, 5)
        f, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg)
        b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg)
        assert np.array_equal(f["tri"], b["tri"]) and np.array_equal(f["t"], b["t"])
