"""Option sweep (default: min_descend_lanes): ms/frame of the three bench workloads for both node layouts.
usage: exp3.py workloads values [option [key=value ...]]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
def workload(wl):
    if wl == "cornell":
        return pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 1920, 1080, 8
    if wl == "default":
        return pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)), pt.make_camera(**pt.DEFAULT_CAMERA), 1920, 1080, 12
    v, i = pt.grid_mesh(int(wl[4:] or 2236), 8.0, 1.0, 1)
    sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
    data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], sky, pt.synthetic_sky(2048, 1024, 0x5EED))
    return data, pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3))), 3840, 2160, 4
thr = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 4, 6, 8]
opt = sys.argv[3] if len(sys.argv) > 3 else "min_descend_lanes"
for kv in sys.argv[4:]:          # fixed options: key=value
    k, v = kv.split("="); t.set_option(k, int(v))
widths = (2, 4)
for wl in sys.argv[1].split(","):
    data, cam, W, H, D = workload(wl)
    s = pt.make_settings(cam, W, H, D)
    acc = t.malloc(W * H * 16)
    for nw in ((2,) if wl == 'cornell' else (4,)):
        sc = pt.Scene(t, data, node_width=nw)
        print(wl, "width", nw, end=" ", flush=True)
        for m in thr:
            t.set_option(opt, m)
            t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
            t.render(sc, s, 1, 4, acc); t.synchronize()
            c = t.counters()
            print(f"{m}:{c.kernel_ms / 4:.2f}ms({c.rays / c.kernel_ms / 1e6:.2f}G)", end=" ", flush=True)
        print(flush=True)
        sc.close()
    t.free(acc)
