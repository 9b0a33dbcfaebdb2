"""refill_lanes x min_descend_lanes sweep on the three bench workloads (GPU box); the shipped defaults are 8 and auto (8 for an
LDS-resident scene, 24 otherwise)."""
import os, sys, time, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cudapathtracer_amd as pt
t = pt.Tracer(0)
def run(name, data, cam, W, H, D, frames, descends):
    sc = pt.Scene(t, data)
    s = pt.make_settings(cam, W, H, D)
    acc = t.malloc(W * H * 16)
    for md, rf in itertools.product(descends, (4, 8, 12, 16, 24)):
        t.set_option("refill_lanes", rf); t.set_option("min_descend_lanes", md)
        t.render(sc, s, 0, 2, acc); t.synchronize(); t.reset_counters()
        t0 = time.perf_counter(); t.render(sc, s, 2, frames, acc); t.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} min_descend {md:2d} refill {rf:2d}: {dt / frames * 1e3:7.3f} ms/frame {t.counters().rays / dt / 1e6:8.0f} Mrays/s", flush=True)
    t.free(acc)
run("cornell", pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 1920, 1080, 8, 64, (4, 8, 16))
run("default", pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)), pt.make_camera(**pt.DEFAULT_CAMERA), 1920, 1080, 12, 24, (16, 24, 32))
v, i = pt.grid_mesh(2236, 8.0, 1.0, 1)
sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], sky, pt.synthetic_sky(2048, 1024, 0x5EED))
run("grid", data, pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3))), 3840, 2160, 4, 24, (16, 24, 32))
