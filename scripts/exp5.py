"""SBVH split alpha sweep (default scene, 2.1 M-triangle grid), 4-wide nodes."""
import sys
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
env = pt.synthetic_sky(2048, 1024)
v, i = pt.grid_mesh(1024, 8.0, 1.0, 1)
sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
grid = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], sky, env)
for name, data, cam, W, H, depth in (("cornell", pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 1920, 1080, 8),
                                     ("default", pt.builtin_scene(1, env), pt.make_camera(**pt.DEFAULT_CAMERA), 1920, 1080, 12),
                                     ("grid2M", grid, pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3))), 3840, 2160, 4)):
    acc = t.malloc(W * H * 16)
    s = pt.make_settings(cam, W, H, depth)
    for alpha in (1e-5, 1e-7, 1e-9, 1e-4, 1e-3, 1e-2, 1.0):
        for rep in range(2):
            sc = pt.Scene(t, data, split_alpha=alpha)
            t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
            t.render(sc, s, 1, 3, acc); t.synchronize()
            c = t.counters()
            n, r = sc.bvh()
            print(f"{name} alpha {alpha:g}: nodes {len(n)} refs {len(r)} {c.kernel_ms/3:7.2f} ms/frame {c.rays/c.kernel_ms/1e3:8.0f} Mrays/s", flush=True)
            sc.close()
    t.free(acc)
