"""Option sweep for the wave-stack integrator on a GPU box."""
import sys, time, itertools
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 1920, 1080
acc = t.malloc(W * H * 16)
for name, which, cam, depth, env in (("cornell", 0, pt.CORNELL_CAMERA, 8, None), ("default", 1, pt.DEFAULT_CAMERA, 12, pt.synthetic_sky(2048, 1024))):
    sc = pt.Scene(t, pt.builtin_scene(which, env))
    s = pt.make_settings(pt.make_camera(**cam), W, H, depth)
    for bpc, refill in itertools.product((4, 8), (1, 4, 8, 16, 32, 64)):
        t.set_option("blocks_per_cu", bpc); t.set_option("refill_lanes", refill)
        t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
        t0 = time.perf_counter(); t.render(sc, s, 1, 4, acc); t.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} blocks/cu {bpc} refill {refill:2d}: {dt/4*1e3:7.2f} ms/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
