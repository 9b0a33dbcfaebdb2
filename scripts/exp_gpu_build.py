#!/usr/bin/env python3
"""GPU BVH build against the host SBVH build (GPU box): build time of mipt_scene_create and trace rate on the result.
usage: python3 scripts/exp_gpu_build.py [grid_n ...]     (grid_n^2 * 2 triangles; 2236 = the bench's 10 M-triangle grid)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MIPT_BUILD_VERBOSE"] = "1"
import cudapathtracer_amd as pt

t = pt.Tracer(0)
sizes = [int(a) for a in sys.argv[1:]] or [300, 1000, 2236]


def measure(name, data, cam, W, H, D):
    for builder, label in ((pt.BVH_LBVH_GPU, "LBVH on the GPU"), (pt.BVH_SBVH, "SBVH on the host")):
        t0 = time.time()
        sc = pt.Scene(t, data, builder=builder)
        build_s = time.time() - t0
        s = pt.make_settings(cam, W, H, D)
        acc = t.malloc(W * H * 16)
        t.render(sc, s, 0, 4, acc); t.synchronize(); t.reset_counters()
        t0 = time.time()
        t.render(sc, s, 4, 32, acc); t.synchronize()
        dt = time.time() - t0
        c = t.counters()
        nodes = len(sc.bvh()[0])
        print(f"{name:>14} {label:>18}: mipt_scene_create {build_s:7.2f} s, {nodes:9d} nodes (width {sc.bvh_width}); "
              f"{c.rays / 32 / 1e6:7.1f} Mrays/frame, {dt / 32 * 1e3:7.2f} ms/frame, {c.rays / dt / 1e9:6.2f} Grays/s", flush=True)
        t.free(acc); sc.close()


sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
measure("default scene", pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)), pt.make_camera(**pt.DEFAULT_CAMERA), 1920, 1080, 12)
for n in sizes:
    v, i = pt.grid_mesh(n, 8.0, 1.0, 1)
    data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], sky, pt.synthetic_sky(2048, 1024, 0x5EED))
    cam = pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3)))
    measure(f"grid {2 * n * n / 1e6:.2f} M", data, cam, 3840, 2160, 4)
