"""GPU parity tests: the HIP path (through the C-ABI of libmipt.so) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): closest-hit primitive id and t bit-exact on fixed ray batches (ties excepted);
images at matched spp within a stated tolerance.  Because both sides evaluate the same fp32 expressions, the
tests ask for more: every hit field bit-exact, per-pixel ray counts identical, radiance within 2e-5 relative
(the only difference left is the order of fp32 additions into a pixel).
"""
import os

import numpy as np
import pytest

from conftest import ROOT, hits_equal, random_rays, soup_scene
from test_oracle_kat import ulp_diff
from test_oracle_render import MG, GOLD, CROP

pytestmark = pytest.mark.gpu

RTOL = 2e-5    # per-pixel radiance, relative to max(|expected|, ATOL_FLOOR)
FLOOR = 1e-3


def rel_err(got, exp):
    return np.abs(got - exp) / np.maximum(np.abs(exp), FLOOR)


def dev_buffers(t, w, h, ldr=True):
    acc = t.malloc(w * h * 16)
    t.memset(acc, 0, w * h * 16)
    l = t.malloc(w * h * 4) if ldr else None
    return acc, l


def test_device_math_bit_exact(tracer, oracle):
    rng = np.random.default_rng(0)
    n = 1 << 18
    cases = [(0, rng.uniform(0, 6.2832, n), None), (1, rng.uniform(0, 6.2832, n), None),
             (2, rng.uniform(-2, 2, n), rng.uniform(-2, 2, n)), (3, rng.uniform(-1, 1, n), None),
             (4, rng.uniform(0, 1, n), None), (5, rng.uniform(0, 2, n), None), (6, rng.uniform(0, 1, n), None),
             (2, np.array([0, 0, 1, -1, 1e-30, 5], np.float32), np.array([0, -1, 0, 0, 1e30, 0], np.float32)),
             (3, np.array([1, -1, 0.5, -0.5, 0, 0.9999999], np.float32), None), (4, np.array([0, 1e-31, 1e-20, 0.5, 1], np.float32), None)]
    for fn, a, b in cases:
        a = np.asarray(a, np.float32)
        b = None if b is None else np.asarray(b, np.float32)
        g = tracer.test_math(fn, a, b)
        c = oracle.math_eval(fn, oracle.MATH_DET, a, b)
        assert np.array_equal(g.view(np.uint32), c.view(np.uint32)), fn


def test_short_reciprocal_equals_ieee_division_for_every_float_in_range(tracer):
    """The integrator computes 1 / d and 1 / det as v_rcp_f32 + one Newton correction (3 instructions); the sweep compares
    that with the compiler's IEEE division for all 3.37 G floats of magnitude 2^-100 .. 2^100 (what unit-scale directions
    and the accepted vertex coordinates can produce).  Zero differences, so the CPU oracle's divisions are matched bit
    for bit; the bare estimate differs on about a tenth of them, which shows that the sweep sees a difference."""
    compared, differing, first, worse = tracer.test_rcp_sweep(1)
    assert compared == 2 * 201 * (1 << 23) and differing == 0 and first == 0 and worse == 0
    c0, d0, _, w0 = tracer.test_rcp_sweep(0)
    assert c0 == compared and d0 > compared // 20 and w0 == 0


def test_short_square_root_equals_sqrtf_for_every_float_in_range(tracer):
    """The direction samplers take their square roots (of a 24-bit draw v, of 1 - v, of 1 - x * x with |x| <= 1: zero or at
    least 2^-24) through sqrt_unit, the compiler's correctly rounded expansion less the rescaling that only inputs below
    2^-96 need.  Compared with sqrtf for zero and all 1.65 G positive floats in 2^-96 .. 2^100: no difference."""
    compared, differing, first, worse = tracer.test_rcp_sweep(4)
    assert compared == 1 + 197 * (1 << 23) and differing == 0 and first == 0 and worse == 0


@pytest.mark.parametrize("width", [2, 4], ids=["bvh2", "q4"])
@pytest.mark.parametrize("builder", [0, 1])
@pytest.mark.parametrize("scene", ["cornell", "default", "soup"])
def test_closest_hit_bit_exact(tracer, ptlib, oracle, builder, scene, width):
    data = {"cornell": lambda: ptlib.builtin_scene(0), "default": lambda: ptlib.builtin_scene(1),
            "soup": lambda: soup_scene(20000, 5)}[scene]()
    sc = ptlib.Scene(tracer, data, builder=builder, node_width=width)
    assert sc.bvh_width == width
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    og, dg = random_rays(300000, 17)
    g = tracer.trace_rays(sc, og, dg)
    c, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
    assert hits_equal(g, c) == 0
    assert (c["tri"] >= 0).sum() > 1000
    # axis-aligned and degenerate directions (idir guard, CUDATracerProg.cu:568-571)
    ax = np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0.6, 0.8, 0), (0, -0.6, 0.8)], np.float32)
    o2 = np.tile(np.array([(0.1, 0.2, 0.3)], np.float32), (len(ax), 1))
    assert hits_equal(tracer.trace_rays(sc, o2, ax), o.trace_rays(oracle.ISECT_FLAT, o2, ax)[0]) == 0
    # empty batch
    assert len(tracer.trace_rays(sc, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0


@pytest.mark.parametrize("width", [2, 4], ids=["bvh2", "q4"])
def test_coincident_and_degenerate_triangles(tracer, ptlib, oracle, width):
    """awkward geometry: a pile of identical triangles (exact t ties: the lowest triangle id wins, on every layout and on
    the brute-force oracle alike), zero-area triangles (|det| < eps: never hit), and a soup around them"""
    from conftest import triangle_soup
    v, i = triangle_soup(400, 9)
    pile = np.tile(v[:3], 60)                                  # 60 copies of the soup's first triangle
    degen = np.tile(v[3:4], 90)                                # 30 triangles collapsed to one point
    line = v[6:9].copy(); line["position"][2] = 0.5 * (line["position"][0] + line["position"][1])     # collinear corners
    verts = np.concatenate([v, pile, degen, line])
    idx = np.concatenate([i, len(v) + np.arange(len(pile) + len(degen) + 3, dtype=np.uint32)])
    data = ptlib.SceneData([(verts, idx, ptlib.make_material())], soup_scene(1, 1).sky)
    sc = ptlib.Scene(tracer, data, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(200000, 23)
    # rays aimed at the pile, so that the ties are actually exercised
    c = verts["position"][len(v):len(v) + 3].mean(axis=0)
    aim = c[None, :] + np.random.default_rng(5).normal(scale=0.05, size=(50000, 3)).astype(np.float32) - og[:50000]
    dg[:50000] = aim / np.linalg.norm(aim, axis=1, keepdims=True)
    g = tracer.trace_rays(sc, og, dg)
    f, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
    b, _ = o.trace_rays(oracle.ISECT_BRUTE, og, dg, nthreads=16)
    assert hits_equal(g, f) == 0
    assert np.array_equal(g["tri"], b["tri"]) and np.array_equal(g["t"].view(np.uint32), b["t"].view(np.uint32))
    first_pile = len(i) // 3
    on_pile = (g["tri"] >= first_pile) & (g["tri"] < first_pile + 60)
    assert on_pile.sum() == 0 and (g["tri"] == 0).sum() > 1000          # the pile's copies tie with triangle 0 and lose to it
    assert (g["tri"] >= first_pile + 60).sum() == 0                       # nothing ever hits a zero-area triangle
    # and the integrator renders it like the oracle does (ray trees per pixel)
    W, H = 96, 64
    s = ptlib.make_settings(ptlib.make_camera((-6.0, 0.5, 0.3), 0.0, 1.5, 1.0), W, H, 4)
    acc, ldr = dev_buffers(tracer, W, H)
    tracer.set_option("count_pixel_rays", 1)
    try:
        tracer.trace(sc, s, acc, ldr)
        rgb, rays = tracer.last_sample(W * H, rays=True)
        ref = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1, acc=np.zeros((H, W, 4), np.float32), nthreads=16)
        assert np.array_equal(rays.reshape(H, W), ref["rays"]) and rays.sum() > 2 * W * H
        assert rel_err(rgb.reshape(H, W, 3), ref["sample"]).max() < RTOL
    finally:
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


@pytest.mark.parametrize("name", ["cornell", "default"])
def test_golden_rays(tracer, ptlib, name):
    # committed fixture: brute-force closest hits from the oracle; the BVH path must reproduce primitive id and t
    g = np.load(os.path.join(GOLD, f"rays_{name}.npz"))
    sc = ptlib.Scene(tracer, MG.scene_data(name))
    h = tracer.trace_rays(sc, g["origins"], g["dirs"])
    exp = g["hits"]
    bad = np.nonzero((h["tri"] != exp["tri"]) | (h["t"].view(np.uint32) != exp["t"].view(np.uint32)))[0]
    assert len(bad) <= 8, len(bad)      # ties excepted: rays that start inside a box see its bottom coplanar with the floor
    assert (ulp_diff(h["t"][bad], exp["t"][bad]) <= 4).all()
    ok = np.setdiff1d(np.arange(len(exp)), bad)
    assert hits_equal(h[ok], exp[ok]) == 0


@pytest.mark.parametrize("name", ["cornell", "default"])
def test_golden_render_crop(tracer, ptlib, oracle, name):
    which, cam_kw, env, depth, frames = MG.CONFIGS[name]
    sc = ptlib.Scene(tracer, MG.scene_data(name))
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), 256, 256, depth)
    acc, ldr = dev_buffers(tracer, 256, 256)
    tracer.render(sc, s, 0, frames, acc, ldr)
    a = tracer.download(acc, (256, 256, 4), np.float32)
    l = tracer.download(ldr, (256, 256, 4), np.uint8)
    tracer.free(acc), tracer.free(ldr)
    g = np.load(os.path.join(GOLD, f"render_{name}_node_det.npz"))
    x0, y0, x1, y1 = CROP
    assert rel_err(a[y0:y1, x0:x1], g["acc"]).max() < RTOL * 4      # 8-16 frames of running mean
    assert np.abs(l[y0:y1, x0:x1].astype(int) - g["ldr"].astype(int)).max() <= 1


@pytest.mark.parametrize("integrator,width", [(0, 2), (1, 2), (0, 4), (1, 4)], ids=["wavestack", "levelqueues", "wavestack-q4", "levelqueues-q4"])
@pytest.mark.parametrize("name,w,h,depth", [("cornell", 160, 96, 5), ("default", 128, 128, 7), ("cornell", 67, 33, 1), ("default", 64, 64, 12),
                                            ("cornell", 1, 1, 3), ("cornell", 5, 3, 14), ("default", 3, 7, 16)])
def test_frame_matches_oracle(tracer, ptlib, oracle, name, w, h, depth, integrator, width):
    which, cam_kw, env, _, _ = MG.CONFIGS[name]
    data = MG.scene_data(name)
    sc = ptlib.Scene(tracer, data, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    tracer.set_option("integrator", integrator)
    tracer.reset_counters()
    try:
        exp_acc = np.zeros((h, w, 4), np.float32)
        for frame in (0, 1, 5):      # running mean continues from whatever acc holds, like the reference
            s.frameID = frame
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, acc=exp_acc, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"]), frame          # identical ray trees
            assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL, frame
            a = tracer.download(acc, (h, w, 4), np.float32)
            assert rel_err(a, r["acc"]).max() < RTOL and (a[..., 3] == 1).all()
            l = tracer.download(ldr, (h, w, 4), np.uint8)
            assert np.abs(l.astype(int) - r["ldr"].astype(int)).max() <= 1 and (l != r["ldr"]).mean() < 1e-3
            assert l[..., :3].max() <= 186 and (l[..., 3] == 255).all()
        assert tracer.counters().frames == 3
    finally:
        tracer.set_option("integrator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_gpu_matches_the_reference_screenshot(tracer, ptlib):
    """The product itself against the one expected output the reference holds (its Cornell-box README screenshot, as 40 x 24
    block means; see test_oracle_render.screenshot_agreement): the app's configuration, 800 x 480, maxDepth 4, 256 spp."""
    from test_oracle_render import screenshot_agreement, check_screenshot_structure
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), 800, 480, 4)
    acc, _ = dev_buffers(tracer, 800, 480, ldr=False)
    tracer.render(sc, s, 0, 256, acc)
    a = tracer.download(acc, (480, 800, 4), np.float32)
    tracer.free(acc)
    stats, m, _ = screenshot_agreement(a)
    for bias, mad, corr in stats:
        assert abs(bias) < 1.5 and mad < 3.0 and corr > 0.99, stats
    check_screenshot_structure(m)


def test_gpu_default_scene_matches_the_reference_screenshot(tracer, ptlib):
    """The product against the reference's other screenshot (Default scene; camera fitted to its outlines, see
    test_oracle_render.test_default_scene_matches_the_reference_screenshot): the closest hits of libmipt's own rays through
    the 200 x 120 pixel centres must cover, per object, the pixels the screenshot's colours assign to it."""
    import importlib.util
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("fit_default_camera", os.path.join(gold, "fit_default_camera.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    cam = json.load(open(os.path.join(gold, "ref_default_camera.json")))
    data = ptlib.builtin_scene(1, None)
    tri_mesh = F.triangle_meshes(data)
    _, masks = F.classes_of_screenshot()
    O, D = F.primary_rays(np.array([*cam["pos"], cam["azimuth"], cam["zenith"]]))
    for width in (2, 4):
        sc = ptlib.Scene(tracer, data, node_width=width)
        t = tracer.trace_rays(sc, O, D)["tri"].reshape(F.H, F.W)
        mm = np.where(t >= 0, tri_mesh[np.maximum(t, 0)], -1)
        ov = F.overlaps(mm, masks)
        assert ov[F.TEAL] > 0.97 and ov[F.GOLD] > 0.85 and ov[F.GREEN] > 0.7 and ov[F.GROUND] > 0.9, ov
        assert (mm == F.GLASS).sum() > 600 and ((mm == F.GLASS) & (masks[F.TEAL] | masks[F.GOLD] | masks[F.GREEN])).sum() == 0


def test_env_map_lookup(tracer, ptlib, oracle):
    # an open scene: every path ends in an env-map texel (OnMiss, CUDATracerProg.cu:809-844), incl. poles and seam
    env = ptlib.synthetic_sky(128, 64, 7)
    v = np.zeros(3, ptlib.VERTEX_DTYPE)
    v["position"] = [(100, 100, 100), (100, 100, 101), (101, 100, 101)]
    v["normal"] = (0, 1, 0)
    sky = ptlib.SkyLight()
    data = ptlib.SceneData([(v, np.arange(3, dtype=np.uint32), ptlib.make_material())], sky, env)
    sc = ptlib.Scene(tracer, data)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    for zen in (0.0, 1.3, float(np.float32(np.pi))):
        cam = ptlib.make_camera((0, 0, 0), 2.0, zen, 1.0)
        s = ptlib.make_settings(cam, 96, 64, 3)
        acc, _ = dev_buffers(tracer, 96, 64, ldr=False)
        tracer.set_option("keep_sample", 1)          # a pipelined frame's fold would clear the sample buffer
        tracer.render(sc, s, 0, 1, acc)
        rgb = tracer.last_sample(96 * 64)
        tracer.set_option("keep_sample", 0)
        tracer.free(acc)
        r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1)
        assert np.array_equal(rgb.reshape(64, 96, 3), r["sample"])     # one contribution per pixel: exact


def test_render_batch_equals_trace_loop(tracer, ptlib):
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), 128, 72, 4)
    acc1, _ = dev_buffers(tracer, 128, 72, ldr=False)
    acc2, _ = dev_buffers(tracer, 128, 72, ldr=False)
    tracer.render(sc, s, 0, 6, acc1)
    for f in range(6):                   # RendererWindow::render loop: Trace, then the CALLER bumps frameID
        s.frameID = f
        tracer.trace(sc, s, acc2, None)
    a1, a2 = tracer.download(acc1, (72, 128, 4), np.float32), tracer.download(acc2, (72, 128, 4), np.float32)
    assert rel_err(a1, a2).max() < RTOL
    # sum mode with a stride: what rank r of G renders; the sum over ranks / N equals the running mean
    total = np.zeros((72, 128, 4), np.float32)
    for r in range(2):
        tracer.memset(acc2, 0, 128 * 72 * 16)
        tracer.render(sc, s, r, 3, acc2, None, frame_stride=2, sum_mode=1)
        total += tracer.download(acc2, (72, 128, 4), np.float32)
    assert rel_err(total[..., :3] / 6, a1[..., :3]).max() < 1e-4
    tracer.free(acc1), tracer.free(acc2)


def test_tile_size_does_not_change_the_image(tracer, ptlib):
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(1, ptlib.synthetic_sky(64, 32)))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.DEFAULT_CAMERA), 200, 120, 6)
    acc, _ = dev_buffers(tracer, 200, 120, ldr=False)
    out = []
    try:
        tracer.set_option("keep_sample", 1)
        tracer.set_option("integrator", 1)          # tiles exist only in the level-synchronous integrator
        for tile in (0, 4096, 777, 64):
            tracer.set_option("tile_pixels", tile)
            tracer.reset_counters()
            tracer.render(sc, s, 3, 1, acc)
            out.append((tracer.last_sample(200 * 120), tracer.counters().rays))
        tracer.set_option("integrator", 0)          # ... and the wave-stack integrator must agree with it
        tracer.reset_counters()
        tracer.render(sc, s, 3, 1, acc)
        out.append((tracer.last_sample(200 * 120), tracer.counters().rays))
    finally:
        tracer.set_option("keep_sample", 0)
        tracer.set_option("integrator", 0)
        tracer.set_option("tile_pixels", 0)
        tracer.free(acc)
    for rgb, rays in out[1:]:
        assert rays == out[0][1] and rel_err(rgb, out[0][0]).max() < RTOL


@pytest.mark.parametrize("name,w,h,depth", [("cornell", 1920, 1080, 8), ("default", 1920, 1080, 12), ("default", 3840, 2160, 4), ("default", 3840, 2160, 12),
                                            ("grid", 3840, 2160, 4), ("grid10m", 3840, 2160, 4)],
                         ids=["C2-cornell-1080p-d8", "C3-default-hdri-1080p-d12", "C4-default-hdri-4k-d4", "C4-default-hdri-4k-d12", "C5-grid-1Mtri-4k-d4", "C5-grid-10Mtri-4k-d4"])
def test_full_size_properties(tracer, ptlib, oracle, name, w, h, depth):
    """BASELINE configs C2 / C3 / C4 / C5 at their full frame sizes (the default scene under the 2048x1024 synthetic HDRI):
    size-independent properties plus an oracle spot check on three rows."""
    if name.startswith("grid"):      # bench.py's C5 stand-in: 2236^2 quads = 10.0 M triangles, and a tenth of it (708^2)
        v, i = ptlib.grid_mesh(2236 if name == "grid10m" else 708, 8.0, 1.0, 1)
        data = ptlib.SceneData([(v, i, ptlib.make_material(color=(0.8, 0.8, 0.8)))], ptlib.SkyLight(), ptlib.synthetic_sky())
        cam_kw = dict(pos=(-7.0, 4.0, 0.0), azimuth=0.0, zenith=2.0, fov=float(np.float32(np.pi / 3)))
    else:
        data = ptlib.builtin_scene(0) if name == "cornell" else ptlib.builtin_scene(1, ptlib.synthetic_sky())
        cam_kw = ptlib.CORNELL_CAMERA if name == "cornell" else ptlib.DEFAULT_CAMERA
    sc = ptlib.Scene(tracer, data)
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    try:
        tracer.reset_counters()
        tracer.render(sc, s, 2, 1, acc, ldr)
        rgb, rays = tracer.last_sample(w * h, rays=True)
        assert tracer.counters().rays == int(rays.sum(dtype=np.uint64))          # conservation: queue totals == per-pixel totals
        assert rays.min() >= 1 and rays.max() <= 2 ** depth - 1 and (rays % 2 == 1).all()   # full binary trees are odd
        assert np.isfinite(rgb).all() and rgb.min() >= 0
        # idempotence: the same frame again gives the same sample (up to atomic-add order)
        tracer.render(sc, s, 2, 1, acc, ldr)
        rgb2, rays2 = tracer.last_sample(w * h, rays=True)
        assert np.array_equal(rays, rays2) and rel_err(rgb2, rgb).max() < RTOL
        l = tracer.download(ldr, (h, w, 4), np.uint8)
        assert l[..., :3].max() <= 186 and (l[..., 3] == 255).all()
        # oracle spot check on three rows of the full-size frame
        o = oracle.Scene(data)
        o.set_flat_bvh(*sc.flat())
        rgb = rgb.reshape(h, w, 3)
        for y in (0, h // 2 - 23, h - 1):
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 2, 1, rect=(0, y, w, y + 1), nthreads=16)
            assert np.array_equal(rays.reshape(h, w)[y], r["rays"][y])
            assert rel_err(rgb[y], r["sample"][y]).max() < RTOL
    finally:
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_tonemap_entry_point(tracer, ptlib, oracle):
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 3, (4096, 4)).astype(np.float32)
    d = tracer.malloc(a.nbytes)
    l = tracer.malloc(4096 * 4)
    tracer.upload(d, a)
    tracer.tonemap(d, 0.5, 4096, l)
    out = tracer.download(l, (4096, 4), np.uint8)
    x = np.clip(a[:, :3] * np.float32(0.5), 0, 1)
    x = (x / (x + 1)).astype(np.float32)
    exp = (oracle.math_eval(4, oracle.MATH_DET, x.reshape(-1)) * np.float32(255)).astype(np.int32).reshape(-1, 3)
    assert np.array_equal(out[:, :3], exp) and (out[:, 3] == 255).all()
    tracer.free(d), tracer.free(l)


def test_error_behaviour(tracer, ptlib):
    import ctypes as C
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    cam = ptlib.make_camera(**ptlib.CORNELL_CAMERA)
    acc, _ = dev_buffers(tracer, 16, 16, ldr=False)
    with pytest.raises(ptlib.MiptError):
        tracer.render(sc, ptlib.make_settings(cam, 0, 16, 4), 0, 1, acc)          # empty viewport
    with pytest.raises(ptlib.MiptError):
        tracer.render(sc, ptlib.make_settings(cam, 16, 16, 31), 0, 1, acc)        # node ids are 32-bit
    with pytest.raises(ptlib.MiptError):
        tracer.render(sc, ptlib.make_settings(cam, 16, 16, 4), 0, 1, None)        # null accumulation buffer
    with pytest.raises(ptlib.MiptError):
        tracer.set_option("no_such_option", 1)
    assert "unknown option" in tracer.last_error()
    host_only = ptlib.Scene(None, ptlib.builtin_scene(0))
    with pytest.raises(ptlib.MiptError):
        tracer.render(host_only, ptlib.make_settings(cam, 16, 16, 4), 0, 1, acc)  # scene of another (no) tracer
    h = C.c_void_p()
    assert ptlib.lib.mipt_create(9999, C.byref(h)) != 0 and not h.value
    # maxDepth 0: nothing traced, black frame
    tracer.reset_counters()
    tracer.render(sc, ptlib.make_settings(cam, 16, 16, 0), 0, 1, acc)
    assert tracer.counters().rays == 0 and not tracer.last_sample(256).any()
    # a pipelined frame's fold consumes its sample buffer: reading it back afterwards is an error, not stale data
    tracer.render(sc, ptlib.make_settings(cam, 16, 16, 2), 0, 1, acc)
    with pytest.raises(ptlib.MiptError, match="keep_sample"):
        tracer.last_sample(256)
    tracer.free(acc)


def test_device_buffer_layer(tracer, ptlib):
    """CUDABuffer semantics (CUDABuffer.cpp:111-224): lazy mirrors, zero-filled first touch, head state machine."""
    import ctypes as C
    lib = ptlib.lib
    b = C.c_void_p()
    assert lib.mipt_buffer_create(tracer.h, 4096, C.byref(b)) == 0
    assert lib.mipt_buffer_head(b) == 0 and lib.mipt_buffer_size(b) == 4096            # UNINITIALIZED
    g = lib.mipt_buffer_mutable_gpu_data(b)
    assert g and lib.mipt_buffer_head(b) == 2                                           # HEAD_AT_GPU
    assert not tracer.download(g, 4096, np.uint8).any()                                 # zero-filled device memory
    tracer.memset(g, 0x5A, 4096)
    cpu = lib.mipt_buffer_cpu_data(b)                                                   # blocking D2H, like bitmap->cpu_data()
    assert lib.mipt_buffer_head(b) == 3                                                 # SYNCED
    assert (np.ctypeslib.as_array(C.cast(cpu, C.POINTER(C.c_uint8)), (4096,)) == 0x5A).all()
    m = lib.mipt_buffer_mutable_cpu_data(b)
    assert lib.mipt_buffer_head(b) == 1                                                 # HEAD_AT_CPU
    np.ctypeslib.as_array(C.cast(m, C.POINTER(C.c_uint8)), (4096,))[:] = 7
    g2 = lib.mipt_buffer_gpu_data(b)                                                    # H2D on demand
    assert g2 == g and lib.mipt_buffer_head(b) == 3 and (tracer.download(g2, 4096, np.uint8) == 7).all()
    lib.mipt_buffer_destroy(b)
    b2 = C.c_void_p()
    lib.mipt_buffer_create(tracer.h, 64, C.byref(b2))
    c = lib.mipt_buffer_cpu_data(b2)
    assert lib.mipt_buffer_head(b2) == 1 and not np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint8)), (64,)).any()
    assert lib.mipt_buffer_async_gpu_push(b2) == 0 and lib.mipt_buffer_head(b2) == 3
    tracer.synchronize()
    lib.mipt_buffer_destroy(b2)


@pytest.mark.parametrize("name,w,h,depth,width", [("cornell", 128, 80, 4, 2), ("default", 96, 96, 8, 2), ("cornell", 64, 48, 8, 2), ("default", 96, 96, 8, 4)])
def test_dfs_order_integrator_is_bit_identical_to_the_oracle(tracer, ptlib, oracle, name, w, h, depth, width):
    """rng_mode = dfs: the reference's single LCG stream per pixel-sample in its depth-first order and its bottom-up
    radiance arithmetic.  No atomics are involved, so the sample, the accumulation buffer and the LDR bytes must equal
    the oracle's DFS mode (same flat BVH, portable math) bit for bit."""
    which, cam_kw, env, _, _ = MG.CONFIGS[name]
    data = MG.scene_data(name)
    sc = ptlib.Scene(tracer, data, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("rng_mode", ptlib.RNG_DFS)
    tracer.set_option("count_pixel_rays", 1)
    try:
        exp_acc = np.zeros((h, w, 4), np.float32)
        for frame in (0, 1, 2):
            s.frameID = frame
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_DFS, oracle.MATH_DET, frame, 1, acc=exp_acc, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"])
            assert np.array_equal(rgb.reshape(h, w, 3).view(np.uint32), r["sample"].view(np.uint32))
            assert np.array_equal(tracer.download(acc, (h, w, 4), np.float32).view(np.uint32), r["acc"].view(np.uint32))
            assert np.array_equal(tracer.download(ldr, (h, w, 4), np.uint8), r["ldr"])
        with pytest.raises(ptlib.MiptError):
            tracer.render(sc, ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, 17), 0, 1, acc)
    finally:
        tracer.set_option("rng_mode", ptlib.RNG_NODE)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def _light_sampling_scene(ptlib, kind):
    rng = np.random.default_rng(3)
    env = (rng.uniform(0.2, 1.0, (32, 64, 3)) ** 2).astype(np.float32)
    env[6:9, 10:14] = 40.0
    env[20:, :5] = 0.0
    if kind == "env":
        return ptlib.builtin_scene(1, env), ptlib.DEFAULT_CAMERA
    if kind == "emitters":
        return ptlib.builtin_scene(0), ptlib.CORNELL_CAMERA
    # both: the default scene under the env map with a small two-sided lamp (one quad) hanging above it
    data = ptlib.builtin_scene(1, env)
    v = np.zeros(4, ptlib.VERTEX_DTYPE)
    v["position"] = [(-0.3, 2.2, -0.3), (-0.3, 2.2, 0.3), (0.3, 2.2, 0.3), (0.3, 2.2, -0.3)]
    v["normal"] = (0, -1, 0)
    lamp = (v, np.array([0, 1, 2, 0, 2, 3], np.uint32), ptlib.make_material(color=(1, 1, 1), emissive=(1.0, 0.9, 0.7), strength=25.0))
    return ptlib.SceneData(data.meshes + [lamp], data.sky, env), ptlib.DEFAULT_CAMERA


@pytest.mark.parametrize("kind,width", [("env", 2), ("env", 4), ("emitters", 0), ("both", 4)])
def test_light_sampling_estimator_matches_oracle(tracer, ptlib, oracle, kind, width):
    """option estimator = 1 (include/mipt.h): env-map importance sampling and/or emissive-triangle sampling.  Identical
    ray trees incl. the occlusion rays, radiance to the usual tolerance, the shadow-ray counter equals the oracle's."""
    data, cam_kw = _light_sampling_scene(ptlib, kind)
    sc = ptlib.Scene(tracer, data, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    o.set_estimator(1)
    assert np.array_equal(sc.env_cdf().view(np.uint32), o.env_cdf().view(np.uint32))
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(sc.emitters(), o.emitters()))
    assert (len(sc.env_cdf()) > 0) == (kind != "emitters") and (len(sc.emitters()[0]) > 0) == (kind != "env")
    w, h, depth = 112, 80, 6
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("estimator", 1)
    tracer.set_option("count_pixel_rays", 1)
    try:
        exp_acc = np.zeros((h, w, 4), np.float32)
        for frame in (0, 1, 7):
            s.frameID = frame
            tracer.reset_counters()
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, acc=exp_acc, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"]), frame
            c = tracer.counters()
            assert c.rays == r["stats"].rays and c.shadow_rays == r["stats"].shadow_rays and c.shadow_rays > 0.05 * c.rays
            assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL, frame
            assert rel_err(tracer.download(acc, (h, w, 4), np.float32), r["acc"]).max() < RTOL
            l = tracer.download(ldr, (h, w, 4), np.uint8)
            assert np.abs(l.astype(int) - r["ldr"].astype(int)).max() <= 1
        # the reference's estimator is untouched by the option's existence
        tracer.set_option("estimator", 0)
        tracer.reset_counters()
        s.frameID = 0
        tracer.trace(sc, s, acc, ldr)
        assert tracer.counters().shadow_rays == 0
        # needs something to sample, node-keyed streams and the wave-stack integrator
        tracer.set_option("estimator", 1)
        dark = ptlib.builtin_scene(1)                      # no env map, no emissive mesh
        with pytest.raises(ptlib.MiptError):
            tracer.trace(ptlib.Scene(tracer, dark), s, acc, ldr)
        tracer.set_option("integrator", 1)
        with pytest.raises(ptlib.MiptError):
            tracer.trace(sc, s, acc, ldr)
        tracer.set_option("integrator", 0)
        with pytest.raises(ptlib.MiptError):
            tracer.set_option("estimator", 4)
    finally:
        tracer.set_option("estimator", 0)
        tracer.set_option("integrator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_large_scene_parallel_build_q4_and_cache_on_gpu(tracer, ptlib, oracle, tmp_path):
    # 131 k triangles: BVH built by spliced child builders, collapsed to 4-wide nodes, written to and reloaded from the
    # cache file; both scenes must trace bit-identically to the oracle walking the same nodes
    v, i = ptlib.grid_mesh(256, 8.0, 1.0, 1)
    data = ptlib.SceneData([(v, i, ptlib.make_material())], ptlib.SkyLight())
    cache = str(tmp_path / "grid.bvh")
    built = ptlib.Scene(tracer, data, bvh_cache=cache)
    loaded = ptlib.Scene(tracer, data, bvh_cache=cache)
    assert built.bvh_width == 4 and not built.bvh_was_cached and loaded.bvh_was_cached
    o = oracle.Scene(data)
    o.set_flat_bvh(*loaded.flat())
    assert o.check_flat_bvh() == (0, "")
    og, dg = random_rays(200000, 5, -7, 7)
    og[:, 1] = 1.5 + np.abs(og[:, 1]) * 0.2
    dg[:, 1] = -np.abs(dg[:, 1]) - 0.05
    dg /= np.linalg.norm(dg, axis=1, keepdims=True)
    c, _ = o.trace_rays(oracle.ISECT_FLAT, og, dg, nthreads=16)
    assert (c["tri"] >= 0).mean() > 0.5
    for sc in (built, loaded):
        assert hits_equal(tracer.trace_rays(sc, og, dg), c) == 0


def test_frame_pipeline_equals_serial_frames(tracer, ptlib):
    """Consecutive frames alternate between two streams / buffer sets (mipt_render's frame pipeline); the running mean
    must come out as with one stream, whatever else the caller queues in between (clears, read-backs, tonemap, a
    single-frame Trace loop, sum mode)."""
    data = ptlib.builtin_scene(1, ptlib.synthetic_sky(64, 32))
    sc = ptlib.Scene(tracer, data)
    w, h = 200, 120
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.DEFAULT_CAMERA), w, h, 6)

    def run(pipeline):
        tracer.set_option("pipeline_frames", pipeline)
        acc, ldr = dev_buffers(tracer, w, h)
        tracer.render(sc, s, 0, 7, acc, ldr)                         # one batch of 7 frames
        a7 = tracer.download(acc, (h, w, 4), np.float32)            # read-back in the middle
        for f in range(7, 12):                                       # Trace loop, one frame per call, no sync in between
            s.frameID = f
            tracer.trace(sc, s, acc, ldr)
        a12 = tracer.download(acc, (h, w, 4), np.float32)
        l12 = tracer.download(ldr, (h, w, 4), np.uint8)
        tracer.memset(acc, 0, w * h * 16)                            # a clear on the tracer's stream, then sum mode with a stride
        tracer.render(sc, s, 1, 5, acc, None, 2, 1)
        tm = tracer.malloc(w * h * 4)
        tracer.tonemap(acc, 1.0 / 5, w * h, tm)
        asum = tracer.download(acc, (h, w, 4), np.float32)
        ltm = tracer.download(tm, (h, w, 4), np.uint8)
        c = tracer.counters()
        tracer.free(acc), tracer.free(ldr), tracer.free(tm)
        return a7, a12, l12, asum, ltm, c.rays, c.frames

    tracer.reset_counters()
    try:
        ser = run(0)
        tracer.reset_counters()
        pip = run(1)
    finally:
        tracer.set_option("pipeline_frames", 1)
        s.frameID = 0
    for k in (0, 1, 3):
        assert rel_err(pip[k], ser[k]).max() < RTOL * 4, k
    for k in (2, 4):
        assert np.abs(pip[k].astype(int) - ser[k].astype(int)).max() <= 1
    assert pip[5] == ser[5] and pip[6] == ser[6] == 17                 # same rays, same frame count


def test_caller_supplied_stream_orders_everything_there(tracer, ptlib):
    """mipt_set_stream: frames and clears run on the caller's stream, in order, so the caller can consume the buffers
    on that stream without mipt_synchronize; the frame pipeline is off.  (A raw HIP stream from the runtime libmipt
    itself is linked against; bench.py does the same with torch's tensors.)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    w, h = 160, 96
    nbytes = w * h * 16
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), w, h, 5)
    acc0 = tracer.malloc(nbytes)
    tracer.memset(acc0, 0, nbytes)
    tracer.render(sc, s, 0, 6, acc0)                                  # reference result on the tracer's own streams
    ref = tracer.download(acc0, (h, w, 4), np.float32)
    acc = tracer.malloc(nbytes)
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    got = np.zeros((h, w, 4), np.float32)
    try:
        tracer.set_stream(st)
        tracer.memset(acc, 0, nbytes)                                  # on the caller's stream now
        tracer.render(sc, s, 0, 6, acc)
        # consumed on the same stream, no library synchronisation in between (2 = hipMemcpyDeviceToHost)
        assert hip.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), C.c_void_p(acc), C.c_size_t(nbytes), 2, st) == 0
        assert hip.hipStreamSynchronize(st) == 0
    finally:
        tracer.set_stream(None)
        hip.hipStreamDestroy(st)
    assert rel_err(got, ref).max() < RTOL * 4
    # back on its own streams the tracer pipelines again and still agrees
    tracer.memset(acc, 0, nbytes)
    tracer.render(sc, s, 0, 6, acc)
    assert rel_err(tracer.download(acc, (h, w, 4), np.float32), ref).max() < RTOL * 4
    tracer.free(acc), tracer.free(acc0)


def test_refit_on_device_equals_a_fresh_scene(tracer, ptlib, oracle):
    """mipt_scene_refit: after the vertices move (and a material turns emissive) the uploaded scene renders what a scene
    created from the new data renders -- same ray trees, and the oracle on the re-fitted BVH agrees as usual."""
    from test_scene_bvh import _moved
    rng = np.random.default_rng(5)
    env = (rng.uniform(0.2, 1.0, (32, 64, 3)) ** 2).astype(np.float32)
    data = ptlib.builtin_scene(1, env)
    w, h, depth = 96, 72, 6
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.DEFAULT_CAMERA), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    try:
        for width in (2, 4):
            sc = ptlib.Scene(tracer, data, node_width=width)
            tracer.trace(sc, s, acc, ldr)                       # frames in flight when the refit arrives
            tracer.render(sc, s, 1, 3, acc, ldr)
            moved = _moved(ptlib, data, 0.2)
            v, i, m = moved.meshes[0]
            lit = ptlib.make_material(color=tuple(m.color), roughness=m.roughness, metallic=m.metallic, opacity=m.opacity, n=m.n,
                                      emissive=(1.0, 0.5, 0.2), strength=6.0)
            moved = ptlib.SceneData([(v, i, lit)] + moved.meshes[1:], moved.sky, moved.env)
            sc.refit(moved)
            fresh = ptlib.Scene(tracer, moved, node_width=width)
            og, dg = random_rays(20000, 12)
            a, b = tracer.trace_rays(sc, og, dg), tracer.trace_rays(fresh, og, dg)
            differ = (a["tri"] != b["tri"]) | (a["t"] != b["t"])
            assert differ.sum() <= 5                            # ties between coplanar triangles only
            o = oracle.Scene(moved)
            o.set_flat_bvh(*sc.flat())
            for est in (0, 1):                                  # 1: the emitter table has grown from none to 12 triangles
                tracer.set_option("estimator", est)
                o.set_estimator(est)
                if est:
                    assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(sc.emitters(), o.emitters()))
                    assert len(sc.emitters()[0]) == len(i) // 3
                for frame in (0, 3):
                    s.frameID = frame
                    tracer.trace(sc, s, acc, ldr)
                    rgb, rays = tracer.last_sample(w * h, rays=True)
                    r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, nthreads=16)
                    assert np.array_equal(rays.reshape(h, w), r["rays"]), (width, est, frame)
                    assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL
                    tracer.trace(fresh, s, acc, ldr)
                    rgb2, rays2 = tracer.last_sample(w * h, rays=True)
                    assert np.array_equal(rays, rays2) and rel_err(rgb, rgb2).max() < RTOL
            tracer.set_option("estimator", 0)
    finally:
        tracer.set_option("estimator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


@pytest.mark.parametrize("est", [2, 3])
@pytest.mark.parametrize("name,width", [("default", 4), ("default", 2), ("cornell", 0)])
def test_estimator_2_and_3_match_oracle(tracer, ptlib, oracle, name, width, est):
    """option estimator = 2 (include/mipt.h): VNDF-sampled specular lobe with G1 weight, cosine-weighted diffuse lobe;
    3: the same with estimator 1's light sampling (occlusion rays) under the cosine lobe's density.  Identical ray trees,
    radiance to the usual tolerance, shadow-ray counter equal to the oracle's; and a different image from estimator 0."""
    which, cam_kw, env, _, _ = MG.CONFIGS[name]
    data = MG.scene_data(name)
    sc = ptlib.Scene(tracer, data, node_width=width)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    o.set_estimator(est)
    w, h, depth = 112, 80, 7
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    try:
        exp_acc = np.zeros((h, w, 4), np.float32)
        for frame in (0, 1, 9):
            s.frameID = frame
            tracer.set_option("estimator", est)
            tracer.reset_counters()
            tracer.trace(sc, s, acc, ldr)
            rgb, rays = tracer.last_sample(w * h, rays=True)
            r = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, frame, 1, acc=exp_acc, nthreads=16)
            assert np.array_equal(rays.reshape(h, w), r["rays"]), frame
            c = tracer.counters()
            assert c.rays == r["stats"].rays and c.shadow_rays == r["stats"].shadow_rays and (c.shadow_rays > 0) == (est == 3)
            assert rel_err(rgb.reshape(h, w, 3), r["sample"]).max() < RTOL, frame
            assert rel_err(tracer.download(acc, (h, w, 4), np.float32), r["acc"]).max() < RTOL
            l = tracer.download(ldr, (h, w, 4), np.uint8)
            assert np.abs(l.astype(int) - r["ldr"].astype(int)).max() <= 1
        tracer.set_option("estimator", 0)
        tracer.trace(sc, s, acc, ldr)
        rgb0, rays0 = tracer.last_sample(w * h, rays=True)
        assert not np.array_equal(rays0, rays) or rel_err(rgb0, rgb).max() > 1e-2      # another estimator, another image
        tracer.set_option("estimator", est)
        tracer.set_option("integrator", 1)
        with pytest.raises(ptlib.MiptError):
            tracer.trace(sc, s, acc, ldr)
    finally:
        tracer.set_option("integrator", 0)
        tracer.set_option("estimator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_tile_partition_renders_each_pixel_as_the_full_frame_does(tracer, ptlib):
    """mipt_set_tile_partition: the parts together are the frame (same ray tree per pixel, radiance to the usual
    tolerance), each part leaves the other parts' pixels alone, and the running mean continues inside the tiles."""
    from cudapathtracer_amd.dist import tiles_for_rank
    data = MG.scene_data("default")
    sc = ptlib.Scene(tracer, data)
    w, h, depth, T, G = 120, 77, 6, 8, 3            # 10 tiles, the last one 5 rows
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.DEFAULT_CAMERA), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    acc2, ldr2 = dev_buffers(tracer, w, h)
    tracer.set_option("count_pixel_rays", 1)
    try:
        for frame in (0, 1, 2):
            s.frameID = frame
            tracer.trace(sc, s, acc, ldr)
        full, full_ldr = tracer.download(acc, (h, w, 4), np.float32), tracer.download(ldr, (h, w, 4), np.uint8)
        _, full_rays = tracer.last_sample(w * h, rays=True)
        total = 0
        for part in range(G):
            tracer.set_tile_partition(T, G, part)
            mine = np.zeros(h, bool)
            for y0, y1 in tiles_for_rank(h, T, part, G):
                mine[y0:y1] = True
            before = tracer.download(acc2, (h, w, 4), np.float32)
            for frame in (0, 1, 2):
                s.frameID = frame
                tracer.reset_counters()
                tracer.trace(sc, s, acc2, ldr2)
            _, rays = tracer.last_sample(w * h, rays=True)
            rays = rays.reshape(h, w)
            assert np.array_equal(rays[mine], full_rays.reshape(h, w)[mine]) and (rays[~mine] == 0).all()
            assert tracer.counters().rays == int(rays.sum(dtype=np.uint64))
            total += int(mine.sum())
            after = tracer.download(acc2, (h, w, 4), np.float32)
            later = np.zeros(h, bool)
            for q in range(part + 1, G):
                for y0, y1 in tiles_for_rank(h, T, q, G):
                    later[y0:y1] = True
            assert np.array_equal(after[later], before[later]) and (after[later] == 0).all()      # not its tiles: untouched
        assert total == h
        got, got_ldr = tracer.download(acc2, (h, w, 4), np.float32), tracer.download(ldr2, (h, w, 4), np.uint8)
        assert rel_err(got, full).max() < RTOL and np.abs(got_ldr.astype(int) - full_ldr.astype(int)).max() <= 1
        # pipelined frames (no per-pixel counter) give the same, and the other integrators refuse a partition
        tracer.set_option("count_pixel_rays", 0)
        tracer.memset(acc2, 0, w * h * 16)
        for part in range(G):
            tracer.set_tile_partition(T, G, part)
            tracer.render(sc, s, 0, 3, acc2, ldr2)
        assert rel_err(tracer.download(acc2, (h, w, 4), np.float32), full).max() < RTOL
        tracer.set_option("integrator", 1)
        with pytest.raises(ptlib.MiptError):
            tracer.trace(sc, s, acc2, ldr2)
        tracer.set_option("integrator", 0)
        with pytest.raises(ptlib.MiptError):
            tracer.set_tile_partition(0, 2, 0)
        with pytest.raises(ptlib.MiptError):
            tracer.set_tile_partition(8, 2, 2)
    finally:
        tracer.set_tile_partition(0, 1, 0)
        tracer.set_option("integrator", 0)
        tracer.set_option("count_pixel_rays", 0)
        for b in (acc, ldr, acc2, ldr2):
            tracer.free(b)


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,depth", [("default", 200, 120, 6), ("cornell", 67, 45, 5), ("default", 5, 3, 4), ("cornell", 1, 1, 3)])
def test_xcd_bands_render_the_same_frame(tracer, ptlib, oracle, name, w, h, depth):
    """Option xcd_bands (automatic for scenes larger than an XCD's L2): the frame is claimed as eight bands, one per XCD
    first, then whatever the others have left.  Who renders a pixel changes nothing about its sample: identical ray
    trees against the oracle, radiance within RTOL -- also for frames with fewer pixels than bands, under a row-tile
    partition, and over pipelined frames."""
    which, cam_kw, env, _, _ = MG.CONFIGS[name]
    data = MG.scene_data(name)
    sc = ptlib.Scene(tracer, data)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    acc, ldr = dev_buffers(tracer, w, h)
    tracer.set_option("xcd_bands", 1)
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
        full = tracer.download(acc, (h, w, 4), np.float32)
        assert rel_err(full, r["acc"]).max() < RTOL
        tracer.set_option("count_pixel_rays", 0)
        s.frameID = 0
        if h >= 8:
            tracer.memset(acc, 0, w * h * 16)
            for part in range(3):
                tracer.set_tile_partition(8, 3, part)
                tracer.render(sc, s, 0, 2, acc, ldr)
            tracer.set_tile_partition(0, 1, 0)
            assert rel_err(tracer.download(acc, (h, w, 4), np.float32), full).max() < RTOL
        tracer.memset(acc, 0, w * h * 16)
        tracer.render(sc, s, 0, 2, acc, ldr)                     # pipelined frames
        assert rel_err(tracer.download(acc, (h, w, 4), np.float32), full).max() < RTOL
        with pytest.raises(ptlib.MiptError):
            tracer.set_option("xcd_bands", 2)
    finally:
        tracer.set_tile_partition(0, 1, 0)
        tracer.set_option("xcd_bands", -1)
        tracer.set_option("count_pixel_rays", 0)
        s.frameID = 0
        tracer.free(acc), tracer.free(ldr)


# ---- wave-stack capacity: a wave that runs out of room stalls, it does not overflow (k_wavestack A1) ----------------------
def test_short_wave_stack_stalls_and_renders_the_same_ray_trees(tracer, ptlib, oracle):
    """test option wave_stack_entries.  The host gives every wave 256 * maxDepth + 256 entries; the capacity sweep
    (profiles/r3/exp_stack_capacity.log) sees the first shade batch cut short below 1024 entries at 1080p, whatever the
    depth.  With 512 entries per wave a full 1080p Cornell frame at depth 8 does run short -- the counter says how often --
    and the waves stall their shade phase instead of overflowing: every pixel still has the oracle's ray tree, synchronize
    reports nothing, and no byte beyond the waves' regions is written."""
    data = ptlib.builtin_scene(0)
    sc = ptlib.Scene(tracer, data)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    W, H = 1920, 1080
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), W, H, 8)
    ref = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1, want_ldr=False, nthreads=16)
    acc, _ = dev_buffers(tracer, W, H, ldr=False)
    tracer.set_option("count_pixel_rays", 1)
    tracer.set_option("wave_stack_entries", 512)
    try:
        tracer.reset_counters()
        tracer.trace(sc, s, acc, None)
        tracer.synchronize()                                   # MIPT_E_LIMIT would raise here
        rgb, rays = tracer.last_sample(W * H, rays=True)
        checked, touched, cut = tracer.test_stack_guard()
        assert cut > 100, cut                                  # the stall path did run
        assert checked > 1 << 20 and touched == 0
        assert np.array_equal(rays.reshape(H, W), ref["rays"])
        assert rel_err(rgb.reshape(H, W, 3), ref["sample"]).max() < 2 * RTOL      # the adds into a pixel come in another order than usual
    finally:
        tracer.set_option("wave_stack_entries", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc)


@pytest.mark.parametrize("name,depth,estimator", [("cornell", 8, 0), ("default", 7, 0), ("cornell", 5, 1)])
def test_tiny_wave_stack_is_exact_or_reported(tracer, ptlib, oracle, name, depth, estimator):
    """far below any sensible capacity (6 ... 300 entries per wave) the stalled waves can reach the one state nothing leaves
    by itself -- stack full, shade queue full, every lane holding a finished ray -- and drop a batch's children; that is
    ALWAYS reported (MIPT_E_LIMIT from mipt_synchronize).  So a frame is either reported or has the oracle's ray tree in
    every pixel; never silently wrong, and never a byte written past the stacks."""
    which, cam_kw, env, _, _ = MG.CONFIGS[name]
    data = MG.scene_data(name)
    sc = ptlib.Scene(tracer, data)
    o = oracle.Scene(data)
    o.set_flat_bvh(*sc.flat())
    o.set_estimator(estimator)
    W, H = 96, 64
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), W, H, depth)
    ref = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1, nthreads=16)
    acc, ldr = dev_buffers(tracer, W, H)
    tracer.set_option("count_pixel_rays", 1)
    tracer.set_option("estimator", estimator)
    exact = 0
    try:
        for cap in (6, 40, 300):
            tracer.set_option("wave_stack_entries", cap)
            tracer.trace(sc, s, acc, ldr)
            try:
                tracer.synchronize()
            except ptlib.MiptError as e:
                assert "-5" in str(e) and "wave stack overflow" in str(e)
                reported = True
            else:
                reported = False
            rgb, rays = tracer.last_sample(W * H, rays=True)
            checked, touched, _ = tracer.test_stack_guard()
            assert checked > 1 << 20 and touched == 0, (cap, checked, touched)
            if reported:
                assert (rays.reshape(H, W) <= ref["rays"]).all()          # rays were lost, none invented
                continue
            exact += 1
            assert np.array_equal(rays.reshape(H, W), ref["rays"]), cap
            assert rel_err(rgb.reshape(H, W, 3), ref["sample"]).max() < 2 * RTOL, cap
        assert exact >= 1
    finally:
        tracer.set_option("wave_stack_entries", 0)
        tracer.set_option("estimator", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_short_wave_stack_under_the_frame_pipeline(tracer, ptlib):
    """the same through mipt_render's pipelined frames (two stack sets, two streams): 8 frames at 720p with 512-entry stacks
    equal 8 frames with the default capacity, ray for ray"""
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    W, H = 1280, 720
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), W, H, 8)
    out = []
    for cap in (0, 512):
        acc, _ = dev_buffers(tracer, W, H, ldr=False)
        tracer.set_option("wave_stack_entries", cap)
        tracer.reset_counters()
        try:
            tracer.render(sc, s, 0, 8, acc)
            tracer.synchronize()
            checked, touched, cut = tracer.test_stack_guard()
            assert (cut > 0) == (cap > 0)
            if cap:
                assert checked > 1 << 20 and touched == 0
        finally:
            tracer.set_option("wave_stack_entries", 0)
        out.append((tracer.download(acc, (H, W, 4), np.float32), tracer.counters().rays))
        tracer.free(acc)
    assert out[1][1] == out[0][1]
    assert rel_err(out[1][0], out[0][0]).max() < RTOL * 4


def test_wave_stack_overflow_is_reported_and_the_next_frame_is_right(tracer, ptlib, oracle):
    """a stack that cannot hold even one record's children (1 entry) always loses rays: the waves still end,
    mipt_synchronize returns MIPT_E_LIMIT once, nothing is written past the stacks, and the next frame at the normal
    capacity is the oracle's again"""
    data = ptlib.builtin_scene(0)
    sc = ptlib.Scene(tracer, data)
    W, H = 96, 64
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), W, H, 8)
    acc, ldr = dev_buffers(tracer, W, H)
    tracer.set_option("count_pixel_rays", 1)
    try:
        tracer.set_option("wave_stack_entries", 1)
        tracer.trace(sc, s, acc, ldr)
        with pytest.raises(ptlib.MiptError, match="-5.*wave stack overflow"):
            tracer.synchronize()
        checked, touched, _ = tracer.test_stack_guard()
        assert checked > 1 << 20 and touched == 0
        _, rays = tracer.last_sample(W * H, rays=True)
        assert rays.max() <= 3 and rays.sum() >= W * H            # primaries (and what idle lanes took over directly); nothing deeper
        tracer.synchronize()                                       # the flag was cleared by the report
        tracer.set_option("wave_stack_entries", 0)
        tracer.memset(acc, 0, W * H * 16)
        tracer.trace(sc, s, acc, ldr)
        tracer.synchronize()
        rgb, rays = tracer.last_sample(W * H, rays=True)
        o = oracle.Scene(data)
        o.set_flat_bvh(*sc.flat())
        ref = o.render(s, oracle.ISECT_FLAT, oracle.RNG_NODE, oracle.MATH_DET, 0, 1, nthreads=16)
        assert np.array_equal(rays.reshape(H, W), ref["rays"]) and rel_err(rgb.reshape(H, W, 3), ref["sample"]).max() < RTOL
    finally:
        tracer.set_option("wave_stack_entries", 0)
        tracer.set_option("count_pixel_rays", 0)
        tracer.free(acc), tracer.free(ldr)


def test_direct_sum_mode_does_not_drift(tracer, ptlib):
    """Sum mode without an LDR target (the multi-GPU path: bench.py with N > 1, libmipt_dist) against the sample-buffer +
    fold path over 2048 frames of a 32 x 32 Default scene under the HDRI, whose contributions span six decades.  With every
    ray added straight into the caller's running sum (round 2) whatever was below half an ulp of that sum was lost:
    -3.7e-5 of the image mean, 2.6e-4 in single pixels (profiles/r3/exp_direct_sum.log).  Blocks of 64 frames keep it to
    rounding noise: stated tolerance 1e-5 on the image mean, 1e-4 per pixel."""
    W = H = 32
    N = 2048
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(1, ptlib.synthetic_sky(256, 128, 0x5EED)))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.DEFAULT_CAMERA), W, H, 12)
    acc, ldr = dev_buffers(tracer, W, H)
    out = []
    for direct in (True, False):
        tracer.memset(acc, 0, W * H * 16)
        done = 0
        for n in (1, 63, 64, 65, 700, N - 893):                     # several calls: blocks stay open across calls and close inside them
            tracer.render(sc, s, done, n, acc, None if direct else ldr, 1, 1)
            done += n
        tracer.synchronize()
        a = tracer.download(acc, (H, W, 4), np.float32)
        assert (a[..., 3] == 1).all()
        out.append(a[..., :3].astype(np.float64) / N)
    d, f = out
    assert abs(d.mean() / f.mean() - 1) < 1e-5, d.mean() / f.mean() - 1
    assert (np.abs(d - f) / np.maximum(np.abs(f), FLOOR)).max() < 1e-4
    assert f.mean() > 1 and f.max() / max(f.min(), 1e-6) > 50      # the wide range the test is about
    tracer.free(acc), tracer.free(ldr)
    sc.close()
