import sys, time
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 1920, 1080
acc = t.malloc(W * H * 16)
def run(name, data, cam, depth, builder=0, leaf=0):
    sc = pt.Scene(t, data, builder=builder, max_leaf=leaf)
    s = pt.make_settings(pt.make_camera(**cam), W, H, depth)
    t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
    t0 = time.perf_counter(); t.render(sc, s, 1, 4, acc); t.synchronize(); dt = time.perf_counter() - t0
    n, r = sc.bvh()
    print(f"{name}: nodes {len(n)} refs {len(r)} {dt/4*1e3:7.2f} ms/frame {t.counters().rays/4/1e6:7.1f} Mrays/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
env = pt.synthetic_sky(2048, 1024)
run("default env d12", pt.builtin_scene(1, env), pt.DEFAULT_CAMERA, 12)
run("default sky d12", pt.builtin_scene(1, None), pt.DEFAULT_CAMERA, 12)
run("default env d8", pt.builtin_scene(1, env), pt.DEFAULT_CAMERA, 8)
run("default env d12 median", pt.builtin_scene(1, env), pt.DEFAULT_CAMERA, 12, builder=1)
for leaf in (1, 2, 8):
    run(f"default env d12 leaf{leaf}", pt.builtin_scene(1, env), pt.DEFAULT_CAMERA, 12, leaf=leaf)
run("cornell d8 median", pt.builtin_scene(0), pt.CORNELL_CAMERA, 8, builder=1)
for leaf in (1, 2, 8):
    run(f"cornell d8 leaf{leaf}", pt.builtin_scene(0), pt.CORNELL_CAMERA, 8, leaf=leaf)
