#!/usr/bin/env python3
"""Emitters first on scenes other than the Cornell box (GPU box): the Default scene under the synthetic HDRI with a one-quad lamp
(lit sky, 4-wide nodes, one-leaf emitter set) and two soups with 600 emitting triangles (an emitters' TREE; black and lit sky).
Each scene is rendered with and without the second tree ($MIPT_EMITTER_TREE=0 at scene creation); same frames."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import cudapathtracer_amd as pt

t = pt.Tracer(0)


def lamp_scene():
    data = pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED))
    v = np.zeros(4, pt.VERTEX_DTYPE)
    v["position"] = [(-0.3, 2.2, -0.3), (-0.3, 2.2, 0.3), (0.3, 2.2, 0.3), (0.3, 2.2, -0.3)]
    v["normal"] = (0, -1, 0)
    lamp = (v, np.array([0, 1, 2, 0, 2, 3], np.uint32), pt.make_material(color=(1, 1, 1), emissive=(1.0, 0.9, 0.7), strength=25.0))
    return pt.SceneData(data.meshes + [lamp], data.sky, data.env), pt.make_camera(**pt.DEFAULT_CAMERA), 12


def soup(lit):
    import test_emitters_first as T
    data = T.emitter_soup(pt, 600, 30000, lit)
    return data, pt.make_camera((-6.0, 0.5, 0.0), 0.0, float(np.float32(np.pi / 2)), float(np.float32(np.pi / 3))), 8


W, H = 1920, 1080
acc = t.malloc(W * H * 16)
for name, (data, cam, depth) in (("default + lamp, HDRI", lamp_scene()), ("soup 600 emitters, black sky", soup(False)), ("soup 600 emitters, lit sky", soup(True))):
    s = pt.make_settings(cam, W, H, depth)
    res = []
    for tree in (True, False):
        os.environ["MIPT_EMITTER_TREE"] = "1" if tree else "0"          # (1: also where the library would not build it by itself)
        sc = pt.Scene(t, data)
        os.environ.pop("MIPT_EMITTER_TREE", None)
        assert bool(sc.emitter_tree()[2]) == tree
        t.render(sc, s, 0, 4, acc); t.synchronize(); t.reset_counters()
        t0 = time.time(); t.render(sc, s, 4, 32, acc); t.synchronize(); dt = time.time() - t0
        c = t.counters()
        res.append((dt / 32 * 1e3, c.rays / 32 / 1e6, sc.bvh_width))
        sc.close()
    (a, ra, w), (b, rb, _) = res
    assert ra == rb
    print(f"{name:>30} (width {w}, depth {depth}): {ra:7.1f} Mrays/frame; emitters first {a:7.2f} ms/frame, closest hit only {b:7.2f} ms/frame -> {b / a:.2f}x", flush=True)
