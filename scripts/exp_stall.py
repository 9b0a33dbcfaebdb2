"""Wave-stack capacity sweep (profiles/r3/exp_stack_capacity.log): per capacity, frames rendered, overflow flags raised,
shade batches cut short for lack of room, Grays/s.  Cornell 1080p/8 and Default 1080p/12, pipelined frames."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W, H = 1920, 1080
acc = t.malloc(W * H * 16)
for name in ("cornell", "default"):
    if name == "cornell":
        sc = pt.Scene(t, pt.builtin_scene(0)); s = pt.make_settings(pt.make_camera(**pt.CORNELL_CAMERA), W, H, 8)
    else:
        sc = pt.Scene(t, pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED))); s = pt.make_settings(pt.make_camera(**pt.DEFAULT_CAMERA), W, H, 12)
    D = s.maxDepth
    for cap in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 128 * D, 256 * D + 256):
        t.set_option("wave_stack_entries", cap)
        flags, frames, rays, sec, short = 0, 0, 0, 0.0, 0
        for rep in range(6):
            t.reset_counters(); t.memset(acc, 0, W * H * 16); t.synchronize()
            t0 = time.perf_counter()
            t.render(sc, s, rep * 16, 16, acc)
            try:
                t.synchronize()
            except pt.MiptError:
                flags += 1
            sec += time.perf_counter() - t0
            frames += 16; rays += t.counters().rays; short += t.test_stack_guard()[2]
        print(f"{name} cap {cap:5d}: frames {frames} flagged batches {flags}/6 cut-short shade batches {short:9d} rays {rays} {rays / sec / 1e9:6.2f} Grays/s", flush=True)
    t.set_option("wave_stack_entries", 0)
    sc.close()
