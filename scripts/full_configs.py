"""BASELINE.json configs C2, C3 and C4 rendered in full on one MI355X (their whole sample counts, not one step):
wall time, spp/s, Grays/s, and a sanity check of the converged image (finite, alpha 1, LDR range).
usage: python scripts/full_configs.py [c2 c3 c4]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt

CONFIGS = {
    "c2": ("Cornell box 1920x1080, 1024 spp, 8 bounces", 0, pt.CORNELL_CAMERA, False, 1920, 1080, 8, 1024),
    "c3": ("DefaultScene + synthetic 2048x1024 HDRI, 1920x1080, 1024 spp, 12 bounces", 1, pt.DEFAULT_CAMERA, True, 1920, 1080, 12, 1024),
    "c4": ("DefaultScene + HDRI 3840x2160, 4096 spp, 12 bounces (one GPU: the whole job a node would split)", 1, pt.DEFAULT_CAMERA, True, 3840, 2160, 12, 4096),
}
t = pt.Tracer(0)
for key in (sys.argv[1:] or ["c2", "c3", "c4"]):
    desc, which, cam, env, W, H, depth, spp = CONFIGS[key]
    sc = pt.Scene(t, pt.builtin_scene(which, pt.synthetic_sky(2048, 1024) if env else None))
    s = pt.make_settings(pt.make_camera(**cam), W, H, depth)
    acc, ldr = t.malloc(W * H * 16), t.malloc(W * H * 4)
    t.memset(acc, 0, W * H * 16)
    t.render(sc, s, 0, 2, acc, ldr); t.synchronize()          # warm-up (allocations); frame 0 restarts the mean below
    t.reset_counters()
    t0 = time.perf_counter()
    chunk = 128
    for f in range(0, spp, chunk):
        t.render(sc, s, f, min(chunk, spp - f), acc, ldr)
        t.synchronize()
        if (f // chunk) % 4 == 3:
            print(f"  {key}: {f + chunk} / {spp} spp, {time.perf_counter() - t0:.1f} s", flush=True)
    dt = time.perf_counter() - t0
    c = t.counters()
    a = t.download(acc, (H, W, 4), np.float32)
    l = t.download(ldr, (H, W, 4), np.uint8)
    ok = bool(np.isfinite(a).all() and (a[..., 3] == 1).all() and l[..., :3].max() <= 186 and (l[..., 3] == 255).all())
    print(f"{key}: {desc}\n  {dt:.2f} s, {spp / dt:.1f} spp/s, {c.rays / dt / 1e9:.2f} Grays/s ({c.rays / 1e9:.1f} G rays), frames {c.frames}, "
          f"mean radiance {a[..., :3].mean():.4f}, image sane: {ok}", flush=True)
    t.free(acc); t.free(ldr)
    del sc
