"""Frame time against frame size (same scene, same depth): the intercept is the per-launch tail of the persistent kernel."""
import sys
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
env = pt.synthetic_sky(2048, 1024)
for name, data, cam, depth in (("cornell", pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 8),
                               ("default", pt.builtin_scene(1, env), pt.make_camera(**pt.DEFAULT_CAMERA), 12)):
    sc = pt.Scene(t, data)
    xs, ys = [], []
    for W, H in ((480, 270), (960, 540), (1920, 1080), (2880, 1620), (3840, 2160)):
        acc = t.malloc(W * H * 16)
        s = pt.make_settings(cam, W, H, depth)
        t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
        t.render(sc, s, 1, 4, acc); t.synchronize()
        c = t.counters()
        ms, mr = c.kernel_ms / 4, c.rays / 4 / 1e6
        xs.append(mr); ys.append(ms)
        print(f"{name} {W}x{H}: {ms:7.3f} ms/frame {mr:8.1f} Mrays/frame {mr / ms:6.2f} Grays/s", flush=True)
        t.free(acc)
    a, b = np.polyfit(xs, ys, 1)
    print(f"{name}: ms = {b:.3f} + Mrays / {1 / a / 1e3:.2f} Grays/s", flush=True)
