#!/usr/bin/env python3
"""Cornell box: SAH triangle cost / leaf size sweep on the current kernels (GPU box).  usage: python3 scripts/exp_cornell_sah.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cudapathtracer_amd as pt
t = pt.Tracer(0)
data, cam, W, H, D = pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 1920, 1080, 8
s = pt.make_settings(cam, W, H, D)
acc = t.malloc(W * H * 16)
for max_leaf in (2, 3, 4, 6, 8):
    for tc in (0.75, 1.0, 1.25, 1.5, 2.0, 3.0):
        try:
            sc = pt.Scene(t, data, tri_cost=tc, max_leaf=max_leaf)
        except pt.MiptError as e:
            print(f"leaf {max_leaf} triCost {tc}: {e}"); continue
        nodes, refs = sc.bvh()
        best = 1e9
        t.render(sc, s, 0, 8, acc); t.synchronize()
        for rep in range(3):
            t.reset_counters()
            t0 = time.time(); t.render(sc, s, 8, 64, acc); t.synchronize(); dt = time.time() - t0
            best = min(best, dt)
        c = t.counters()
        print(f"leaf<={max_leaf} triCost {tc:4.2f}: {len(nodes):3d} nodes {len(refs):3d} refs width {sc.bvh_width} lds_slots {len(nodes) * 4 + len(refs) * 3:3d}  {best / 64 * 1e3:6.3f} ms/frame {c.rays / 64 / best / 1e9:6.2f} Grays/s", flush=True)
        sc.close()
