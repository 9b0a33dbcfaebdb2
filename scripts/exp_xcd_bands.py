"""xcd_bands (one band of the frame per XCD, claimed from band blockIdx % 8 first) off/on on the three bench
workloads (GPU box).  usage: exp_xcd_bands.py [cornell] [default] [grid] [grid:<n>]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
which = sys.argv[1:] or ["cornell", "default", "grid"]
t = pt.Tracer(0)
fov = float(np.float32(np.pi / 3))
for w in which:
    if w.startswith("cornell"):                 # "cornell4k" / "default4k": the same scenes at 3840x2160
        W, H, D = (3840, 2160, 8) if w.endswith("4k") else (1920, 1080, 8)
        data = pt.builtin_scene(0); cam = pt.make_camera(**pt.CORNELL_CAMERA)
    elif w.startswith("default"):
        W, H, D = (3840, 2160, 12) if w.endswith("4k") else (1920, 1080, 12)
        data = pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)); cam = pt.make_camera(**pt.DEFAULT_CAMERA)
    else:                                   # "grid" or "grid:<n>" (n x n quads)
        W, H, D = 3840, 2160, 4
        v, i = pt.grid_mesh(int(w.split(":")[1]) if ":" in w else 2236, 8.0, 1.0, 1)
        data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], pt.SkyLight(), pt.synthetic_sky(2048, 1024))
        cam = pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, fov)
    sc = pt.Scene(t, data)
    acc = t.malloc(W * H * 16)
    s = pt.make_settings(cam, W, H, D)
    for rep in range(2):
        for xb in (0, 1):
            t.set_option("xcd_bands", xb)
            t.render(sc, s, 0, 4, acc); t.synchronize(); t.reset_counters()
            t0 = time.perf_counter(); t.render(sc, s, 4, 16, acc); t.synchronize(); dt = time.perf_counter() - t0
            print(f"{w:9s} xcd_bands {xb}: {dt/16*1e3:7.3f} ms/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
    t.set_option("xcd_bands", -1)
    t.free(acc)
