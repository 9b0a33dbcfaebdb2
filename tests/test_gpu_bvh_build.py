"""GPU-side BVH build (bvhBuilder = MIPT_BVH_LBVH_GPU, csrc/gpu_bvh.hip).  No counterpart in the reference, which
builds on the host (CUDATracer/Types.hpp:312-413).  The bar is the same as for every other tree: closest hits do not depend on
the tree, so the product on the GPU-built tree must agree bit for bit with the oracle traversing the same bytes, the tree must
pass the oracle's own structural check, and a frame must show the oracle's ray trees.
"""
import numpy as np
import pytest

from conftest import hits_equal, random_rays, soup_scene
from test_gpu_parity import RTOL, dev_buffers, rel_err
from test_oracle_render import MG

pytestmark = pytest.mark.gpu


def _scene(ptlib, name):
    if name == "soup":
        return soup_scene(20000, 5)
    if name == "grid":
        v, i = ptlib.grid_mesh(300, 8.0, 1.0, 1)              # 180 k triangles, displaced height field
        sky = ptlib.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
        return ptlib.SceneData([(v, i, ptlib.make_material(color=(0.8, 0.8, 0.8)))], sky)
    return ptlib.builtin_scene(1)


@pytest.mark.parametrize("width", [2, 4, 0], ids=["bvh2", "q4", "auto"])
@pytest.mark.parametrize("scene", ["default", "soup", "grid"])
def test_gpu_built_tree_gives_the_oracles_closest_hits(tracer, ptlib, oracle, scene, width):
    data = _scene(ptlib, scene)
    sc = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, node_width=width)
    assert sc.bvh_width == (width or 4)
    nodes, refs, w = sc.flat()
    ntri = sum(len(m[1]) // 3 for m in data.meshes)
    ids = np.ascontiguousarray(refs[:, 3]).view(np.uint32)          # reference record = v0 | triangle id, e1 | mesh, e2 | flags
    assert len(ids) == ntri and np.array_equal(np.sort(ids), np.arange(ntri, dtype=np.uint32))        # every triangle exactly once: no duplicates
    o = oracle.Scene(data)
    o.set_flat_bvh(nodes, refs, w)
    assert o.check_flat_bvh() == (0, "")
    lo, hi = (-3.0, 3.0) if scene != "grid" else (-5.0, 5.0)
    og, dg = random_rays(300000, 17, lo, hi)
    g = tracer.trace_rays(sc, og, dg)
    c, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
    assert hits_equal(g, c) == 0
    assert (c["tri"] >= 0).sum() > 1000
    # the same rays on the host-built SBVH: the closest hit does not depend on the tree (knife-edge rays aside: a ray through an
    # edge shared by two boxes can be culled by one tree's rounding and not the other's)
    ref = ptlib.Scene(tracer, data, node_width=width)
    r = tracer.trace_rays(ref, og, dg)
    assert hits_equal(g, r) <= 3


@pytest.mark.parametrize("n,max_leaf,width", [(64, 4, 4), (65, 1, 2), (1000, 15, 4), (777, 2, 2), (4097, 1, 4), (300, 8, 0)])
def test_gpu_build_edge_sizes_and_leaf_sizes(tracer, ptlib, oracle, n, max_leaf, width):
    """the smallest scene the builder takes, odd counts, one triangle per leaf and the largest leaves; also coordinates far from
    the origin and a scene squeezed into a sliver (Morton cells are cubes of the largest extent: two axes collapse to one cell)"""
    from conftest import triangle_soup
    for shape in ("soup", "far", "sliver"):
        v, i = triangle_soup(n, n)
        if shape == "far":
            v["position"] += np.float32(3.0e5)
        elif shape == "sliver":
            v["position"][:, 1:] *= np.float32(1e-4)
        data = ptlib.SceneData([(v, i, ptlib.make_material())], soup_scene(1, 1).sky)
        sc = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, max_leaf=max_leaf, node_width=width)
        nodes, refs, w = sc.flat()
        assert len(refs) == n
        o = oracle.Scene(data)
        o.set_flat_bvh(nodes, refs, w)
        assert o.check_flat_bvh() == (0, ""), shape
        og, dg = random_rays(60000, 3)
        if shape == "far":
            og += np.float32(3.0e5)
        elif shape == "sliver":
            og[:, 1:] *= np.float32(1e-4); dg[:, 0] *= np.float32(1e-3); dg /= np.linalg.norm(dg, axis=1, keepdims=True)
        g = tracer.trace_rays(sc, og, dg)
        c, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
        assert hits_equal(g, c) == 0, shape
        assert (c["tri"] >= 0).sum() > 50, shape


def test_gpu_build_is_deterministic_and_refuses_what_it_cannot_do(tracer, ptlib):
    data = soup_scene(50000, 11)
    a = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU).flat()
    b = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU).flat()
    assert a[2] == b[2] == 4 and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    with pytest.raises(ptlib.MiptError, match="fewer than 64 triangles"):
        ptlib.Scene(tracer, ptlib.builtin_scene(0), builder=ptlib.BVH_LBVH_GPU)           # the Cornell box: 36 triangles
    with pytest.raises(ptlib.MiptError):
        ptlib.Scene(None, data, builder=ptlib.BVH_LBVH_GPU)                                # host-only scene: nothing to build on


def test_gpu_build_of_coincident_and_degenerate_triangles(tracer, ptlib, oracle):
    """equal Morton codes by the hundred (a pile of identical triangles, triangles collapsed to a point): the radix tree tells
    them apart by position, the tree stays shallow enough for the traversal stack, exact t ties go to the lowest triangle id"""
    from conftest import triangle_soup
    v, i = triangle_soup(400, 9)
    pile = np.tile(v[:3], 300)
    degen = np.tile(v[3:4], 90)
    verts = np.concatenate([v, pile, degen])
    idx = np.concatenate([i, len(v) + np.arange(len(pile) + len(degen), dtype=np.uint32)])
    data = ptlib.SceneData([(verts, idx, ptlib.make_material())], soup_scene(1, 1).sky)
    for width in (2, 4):
        sc = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, node_width=width)
        o = oracle.Scene(data)
        o.set_flat_bvh(*sc.flat())
        assert o.check_flat_bvh() == (0, "")
        og, dg = random_rays(100000, 23)
        c = verts["position"][len(v):len(v) + 3].mean(axis=0)
        aim = c[None, :] + np.random.default_rng(5).normal(scale=0.05, size=(30000, 3)).astype(np.float32) - og[:30000]
        dg[:30000] = aim / np.linalg.norm(aim, axis=1, keepdims=True)
        g = tracer.trace_rays(sc, og, dg)
        f, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
        assert hits_equal(g, f) == 0
        b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg, nthreads=16)
        assert (g["tri"] != b["tri"]).sum() <= 3                                           # knife-edge rays only
        # the pile is 300 copies of triangle 0: every hit on it is an exact t tie of 301 triangles and goes to the lowest id
        assert (g["tri"] == 0).sum() > 1000 and not ((g["tri"] >= 400) & (g["tri"] < 700)).any()


@pytest.mark.parametrize("width", [2, 4], ids=["bvh2", "q4"])
def test_frame_on_a_gpu_built_tree_matches_the_oracle(tracer, ptlib, oracle, width):
    which, cam_kw, env, _, _ = MG.CONFIGS["default"]
    data = MG.scene_data("default")
    sc = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    w, h, depth = 96, 64, 8
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    try:
        exp_acc = np.zeros((h, w, 4), np.float32)
        for frame in (0, 1):
            s.frameID = frame
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, acc=exp_acc, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"]), frame
            assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL, frame
    finally:
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_gpu_built_tree_can_be_refitted_and_cached(tracer, ptlib, oracle, tmp_path):
    data = soup_scene(30000, 3)
    cache = str(tmp_path / "lbvh.cache")
    sc = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, bvh_cache=cache)
    assert not sc.bvh_was_cached
    again = ptlib.Scene(tracer, data, builder=ptlib.BVH_LBVH_GPU, bvh_cache=cache)
    assert again.bvh_was_cached and np.array_equal(again.flat()[0].view(np.uint32), sc.flat()[0].view(np.uint32))
    # a host-built tree of the same triangles has another cache key
    assert not ptlib.Scene(tracer, data, bvh_cache=cache).bvh_was_cached
    moved_v = data.meshes[0][0].copy()
    moved_v["position"] += np.float32(0.05) * np.sin(np.arange(len(moved_v), dtype=np.float32))[:, None]
    moved = ptlib.SceneData([(moved_v, data.meshes[0][1], data.meshes[0][2])], data.sky)
    sc.refit(moved)
    o = oracle.Scene(moved)
    o.set_flat_bvh(*sc.flat())
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(100000, 31)
    assert hits_equal(tracer.trace_rays(sc, og, dg), o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)[0]) == 0
