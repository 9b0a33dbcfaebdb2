import sys, time, itertools
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 1920, 1080
acc = t.malloc(W * H * 16)
env = pt.synthetic_sky(2048, 1024)
for name, data, cam, depth in (("cornell", pt.builtin_scene(0), pt.CORNELL_CAMERA, 8), ("default", pt.builtin_scene(1, env), pt.DEFAULT_CAMERA, 12)):
    s = pt.make_settings(pt.make_camera(**cam), W, H, depth)
    for leaf, tc in itertools.product((2, 3, 4, 6, 8), (0.5, 1.0, 1.5, 2.0, 3.0)):
        sc = pt.Scene(t, data, max_leaf=leaf, tri_cost=tc)
        t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
        t0 = time.perf_counter(); t.render(sc, s, 1, 3, acc); t.synchronize(); dt = time.perf_counter() - t0
        n, r = sc.bvh()
        print(f"{name} leaf {leaf} triCost {tc}: nodes {len(n)} refs {len(r)} {dt/3*1e3:7.2f} ms/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
        sc.close()
