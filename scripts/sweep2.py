"""refill_lanes x min_descend_lanes sweep of the wave-stack integrator with the frame pipeline on (GPU box)."""
import sys, time, itertools
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 1920, 1080
acc = t.malloc(W * H * 16)
for name, which, cam, depth, env, mds in (("cornell", 0, pt.CORNELL_CAMERA, 8, None, (6, 8, 10)),
                                          ("default", 1, pt.DEFAULT_CAMERA, 12, pt.synthetic_sky(2048, 1024), (20, 22, 24, 26, 28))):
    sc = pt.Scene(t, pt.builtin_scene(which, env))
    s = pt.make_settings(pt.make_camera(**cam), W, H, depth)
    for refill, md in itertools.product((8,), mds):
        t.set_option("refill_lanes", refill); t.set_option("min_descend_lanes", md)
        t.render(sc, s, 0, 2, acc); t.synchronize(); t.reset_counters()
        t0 = time.perf_counter(); t.render(sc, s, 2, 12, acc); t.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} refill {refill:2d} min_descend {md:2d}: {dt/12*1e3:7.3f} ms/frame {t.counters().rays/dt/1e6:8.0f} Mrays/s", flush=True)
