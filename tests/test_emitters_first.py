""""Emitters first" (include/mipt.h mipt_scene_get_emitter_tree, kernels.hip phase D; no counterpart in the reference, which traces
every ray to its closest hit, CUDATracerProg.cu:543-794).  A last-level ray can only add emission or the sky, so the product
intersects it with a small tree over the emitting triangles first and then walks the scene only as far as it takes to know whether
that emitter is the closest hit.  The bar: the image and the per-pixel ray counts are those of the closest-hit algorithm -- on the
CPU the oracle's restatement of the rule is bit-identical to its closest-hit render, on the GPU the product equals the oracle as in
every other parity test -- for one-leaf and many-triangle emitter sets, both node layouts, black and lit skies, and across refits.
"""
import os

import numpy as np
import pytest

from conftest import soup_scene, triangle_soup


def emitter_soup(ptlib, n_emit, n_plain, lit_sky, seed=7):
    """two soups in the same volume: one mesh emits, one does not (and a third, glassy one, so that refraction is in the paths)"""
    ve, ie = triangle_soup(n_emit, seed, extent=2.5, size=0.25)
    vp, ip = triangle_soup(n_plain, seed + 1, extent=2.5, size=0.5)
    vg, ig = triangle_soup(max(n_plain // 4, 1), seed + 2, extent=2.5, size=0.5)
    sky = ptlib.SkyLight()
    sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.57735026, -0.57735026, -0.57735026); sky.intensity = 5.0 if lit_sky else 0.0
    return ptlib.SceneData([(vp, ip, ptlib.make_material(color=(0.7, 0.6, 0.5), roughness=0.4)),
                            (ve, ie, ptlib.make_material(color=(0.9, 0.9, 0.9), emissive=(1.0, 0.8, 0.5), strength=4.0)),
                            (vg, ig, ptlib.make_material(color=(0.8, 0.9, 0.8), opacity=0.3, n=1.4))], sky)


def scene_for(ptlib, name):
    if name == "cornell":
        return ptlib.builtin_scene(0), ptlib.CORNELL_CAMERA
    cam = dict(pos=(-6.0, 0.5, 0.0), azimuth=0.0, zenith=float(np.float32(np.pi / 2)), fov=float(np.float32(np.pi / 3)))
    if name == "lamp-under-env":                                  # the Default scene under an env map with a one-quad lamp
        from test_gpu_parity import _light_sampling_scene
        return _light_sampling_scene(ptlib, "both")
    n_emit, lit = {"few-black": (9, False), "many-black": (600, False), "many-lit": (600, True), "few-lit": (5, True)}[name]
    return emitter_soup(ptlib, n_emit, 3000, lit), cam


SCENES = ["cornell", "few-black", "many-black", "many-lit", "few-lit", "lamp-under-env"]
# Under a lit sky a scene gets the second tree by itself only when its emitters are a tree of their own (api.hip emit_tree_wanted:
# with a handful of emitting triangles the extra leaf per ray costs more than "first hit ends it" saves); the rule is correct
# there all the same, so these two scenes are built with $MIPT_EMITTER_TREE=1 and tested like the others.
FORCED = {"few-lit", "lamp-under-env"}


def make_scene(ptlib, tracer, data, name, **kw):
    if name in FORCED:
        os.environ["MIPT_EMITTER_TREE"] = "1"
    try:
        return ptlib.Scene(tracer, data, **kw)
    finally:
        os.environ.pop("MIPT_EMITTER_TREE", None)


@pytest.mark.parametrize("name", SCENES)
def test_which_scenes_get_the_second_tree_by_themselves(ptlib, name):
    data, _ = scene_for(ptlib, name)
    assert (ptlib.Scene(None, data).emitter_tree()[2] != 0) == (name not in FORCED)


# ---- CPU: layout of the second tree, and the oracle's restatement of the rule ----------------------------------------------------
@pytest.mark.parametrize("width", [2, 4])
@pytest.mark.parametrize("name", SCENES)
def test_emitters_tree_lies_behind_the_main_tree(ptlib, oracle, name, width):
    data, _ = scene_for(ptlib, name)
    sc = make_scene(ptlib, None, data, name, node_width=width)
    nodes, refs, w = sc.flat()
    en, er, root = sc.emitter_tree()
    assert root != 0 and w == width
    emit_ids = np.ascontiguousarray(er[:, 3]).view(np.uint32)
    # exactly the triangles of emitting meshes (the SBVH may reference one from several leaves), flagged as emitting
    base, want = 0, []
    for v, i, m in data.meshes:
        if m.emissiveStrength != 0 and any(c != 0 for c in m.emissiveColor):
            want += list(range(base, base + len(i) // 3))
        base += len(i) // 3
    assert sorted(set(emit_ids.tolist())) == want
    assert (np.ascontiguousarray(er[:, 11]).view(np.uint32) == 2).all()
    if len(want) <= 15:
        assert len(en) == 0 and root == (0x80000000 | (len(want) << 27) | len(refs))       # one leaf, right behind the main references
    else:
        assert root == len(nodes) and len(en) > 0
        # the appended nodes only point into the appended part
        fan = 4 if w == 4 else 2
        r = np.ascontiguousarray(en[:, 12:12 + fan]).view(np.uint32)
        inner = (r & 0x80000000) == 0
        assert (r[inner] >= len(nodes)).all() and (r[inner] < len(nodes) + len(en)).all()
        leaf = ~inner & (((r >> 27) & 0xF) > 0)
        first, cnt = r[leaf] & 0x07FFFFFF, (r[leaf] >> 27) & 0xF
        assert (first >= len(refs)).all() and (first + cnt <= len(refs) + len(er)).all()
    # the oracle accepts it only where it lies
    o = oracle.Scene(data)
    fn, fr, fw, froot, nmain = sc.flat_with_emitters()
    o.set_flat_bvh(fn, fr, fw)
    o.set_emitter_tree(froot, nmain)
    with pytest.raises(ValueError):
        o.set_emitter_tree(len(fn) + 5, nmain)


def test_no_emitters_tree_when_nothing_or_everything_emits(ptlib):
    assert ptlib.Scene(None, ptlib.builtin_scene(1)).emitter_tree()[2] == 0                  # Default scene: no mesh emits
    v, i = triangle_soup(100, 3)
    sky = soup_scene(1, 1).sky
    glow = ptlib.make_material(emissive=(1, 1, 1), strength=2.0)
    assert ptlib.Scene(None, ptlib.SceneData([(v, i, glow)], sky)).emitter_tree()[2] == 0    # everything emits: nothing to tell apart
    os.environ["MIPT_EMITTER_TREE"] = "0"
    try:
        assert ptlib.Scene(None, ptlib.builtin_scene(0)).emitter_tree()[2] == 0
    finally:
        del os.environ["MIPT_EMITTER_TREE"]
    assert ptlib.Scene(None, ptlib.builtin_scene(0)).emitter_tree()[2] != 0


@pytest.mark.parametrize("width", [2, 4])
@pytest.mark.parametrize("name", SCENES)
def test_oracle_emitters_first_is_the_closest_hit_image_with_less_work(ptlib, oracle, name, width):
    data, cam = scene_for(ptlib, name)
    sc = make_scene(ptlib, None, data, name, node_width=width)
    fn, fr, fw, root, nmain = sc.flat_with_emitters()
    s = ptlib.make_settings(ptlib.make_camera(**cam), 72, 48, 5)
    for est in (0, 2):
        o = oracle.Scene(data)
        o.set_flat_bvh(fn, fr, fw)
        o.set_estimator(est)
        a = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 2, nthreads=8)
        o.set_anyhit_last_level(True)
        o.set_emitter_tree(root, nmain)
        b = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 2, nthreads=8)
        assert np.array_equal(a["acc"].view(np.uint32), b["acc"].view(np.uint32)), est       # bit for bit
        assert np.array_equal(a["rays"], b["rays"]) and np.array_equal(a["ldr"], b["ldr"])
        sa, sb = a["stats"], b["stats"]
        assert sa.rays == sb.rays and sb.emitfirst_rays > 0.2 * sb.rays and sa.emitfirst_rays == 0
        assert sb.node_visits < sa.node_visits and sb.tri_tests < 1.2 * sa.tri_tests
        assert a["acc"][..., :3].max() > 0
    # the light-sampling estimators keep their own route (their last-level rays carry flags the rule does not know)
    o.set_estimator(1)
    assert o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1, nthreads=8)["stats"].emitfirst_rays == 0


# ---- GPU: the product against the oracle's closest-hit render ------------------------------------------------------------------------
def _frames_match(tracer, ptlib, oracle, sc, data, cam, w, h, depth, frames=(0, 2), est=0):
    from test_gpu_parity import RTOL, dev_buffers, rel_err
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    o.set_estimator(est)
    s = ptlib.make_settings(ptlib.make_camera(**cam), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    tracer.set_option("estimator", est)
    try:
        out = []
        for frame in frames:
            s.frameID = frame
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"]), frame                      # identical ray trees
            assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL, frame
            out.append((rgb.copy(), rays.copy()))
        return out
    finally:
        tracer.set_option("estimator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


@pytest.mark.gpu
@pytest.mark.parametrize("width", [2, 4])
@pytest.mark.parametrize("name", SCENES)
def test_gpu_emitters_first_matches_the_oracles_closest_hit_render(tracer, ptlib, oracle, name, width):
    from test_gpu_parity import RTOL, rel_err
    data, cam = scene_for(ptlib, name)
    sc = make_scene(ptlib, tracer, data, name, node_width=width)
    assert sc.emitter_tree()[2] != 0
    for depth in (1, 2, 6):                      # depth 1: the primary rays are the last level; 2: every shade batch's children are
        with_tree = _frames_match(tracer, ptlib, oracle, sc, data, cam, 80, 56, depth)
    _frames_match(tracer, ptlib, oracle, sc, data, cam, 64, 40, 5, est=2)
    # and against the product itself without the second tree
    os.environ["MIPT_EMITTER_TREE"] = "0"
    try:
        plain = ptlib.Scene(tracer, data, node_width=width)
    finally:
        del os.environ["MIPT_EMITTER_TREE"]
    assert plain.emitter_tree()[2] == 0
    without = _frames_match(tracer, ptlib, oracle, plain, data, cam, 80, 56, 6)
    for (rgb_a, rays_a), (rgb_b, rays_b) in zip(with_tree, without):
        assert np.array_equal(rays_a, rays_b) and rel_err(rgb_a, rgb_b).max() < 2 * RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("width", [2, 4])
def test_gpu_emitters_first_across_refits(tracer, ptlib, oracle, width):
    """moved vertices: the second tree keeps its shape and is re-fitted; a material edit that changes which triangles emit: the
    tree is rebuilt if it fits the room it had, else the scene goes on without one -- the image is the oracle's either way"""
    data, cam = scene_for(ptlib, "many-black")
    sc = ptlib.Scene(tracer, data, node_width=width)
    root0 = sc.emitter_tree()[2]
    assert root0 != 0
    moved_meshes = []
    for k, (v, i, m) in enumerate(data.meshes):
        v = v.copy()
        v["position"] += np.float32(0.04) * np.sin(np.arange(len(v), dtype=np.float32) + k)[:, None]
        moved_meshes.append((v, i, m))
    moved = ptlib.SceneData(moved_meshes, data.sky)
    sc.refit(moved)
    assert sc.emitter_tree()[2] == root0
    _frames_match(tracer, ptlib, oracle, sc, moved, cam, 72, 48, 5)
    # the plain mesh starts to glow as well: more emitting triangles than the room behind the main tree holds
    v, i, m = moved_meshes[0]
    glow = ptlib.make_material(color=tuple(m.color), roughness=m.roughness, emissive=(0.2, 0.3, 1.0), strength=1.5)
    edited = ptlib.SceneData([(v, i, glow)] + moved_meshes[1:], data.sky)
    sc.refit(edited)
    assert sc.emitter_tree()[2] == 0
    _frames_match(tracer, ptlib, oracle, sc, edited, cam, 72, 48, 5)
    # back to the first set: the tree returns
    sc.refit(moved)
    assert sc.emitter_tree()[2] == root0
    _frames_match(tracer, ptlib, oracle, sc, moved, cam, 72, 48, 5)
    # the lamp goes out: nothing emits
    dark = ptlib.SceneData([moved_meshes[0], (moved_meshes[1][0], moved_meshes[1][1], ptlib.make_material(color=(0.9, 0.9, 0.9))), moved_meshes[2]], data.sky)
    sc.refit(dark)
    assert sc.emitter_tree()[2] == 0
