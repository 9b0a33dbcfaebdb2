"""refill_lanes x blocks_per_cu sweep on the 10 M-triangle grid at 4K / 4 bounces with xcd_bands (GPU box)."""
import sys, time, itertools
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 3840, 2160
acc = t.malloc(W * H * 16)
v, i = pt.grid_mesh(2236, 8.0, 1.0, 1)
data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], pt.SkyLight(), pt.synthetic_sky(2048, 1024))
sc = pt.Scene(t, data)
print("scene built", flush=True)
s = pt.make_settings(pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3))), W, H, 4)
for bpc, refill in itertools.product((4, 3, 2), (4, 8, 16, 32)):
    t.set_option("blocks_per_cu", bpc); t.set_option("refill_lanes", refill)
    t.render(sc, s, 0, 2, acc); t.synchronize(); t.reset_counters()
    t0 = time.perf_counter(); t.render(sc, s, 2, 8, acc); t.synchronize(); dt = time.perf_counter() - t0
    print(f"grid blocks_per_cu {bpc} refill {refill:2d}: {dt/8*1e3:7.3f} ms/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
