"""First-light check on a GPU box: device math, closest hit and one rendered frame against the oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cudapathtracer_amd as pt  # noqa: E402
from oracle import orc  # noqa: E402

t = pt.Tracer(0)
print("tracer ok")

# --- device math bit-parity
rng = np.random.default_rng(0)
for fn, a, b in [(0, rng.uniform(0, 6.2832, 100000), None), (1, rng.uniform(0, 6.2832, 100000), None),
                 (2, rng.uniform(-2, 2, 100000), rng.uniform(-2, 2, 100000)), (3, rng.uniform(-1, 1, 100000), None),
                 (4, rng.uniform(0, 1, 100000), None), (5, rng.uniform(0, 2, 100000), None), (6, rng.uniform(0, 1, 100000), None)]:
    a = a.astype(np.float32)
    b = None if b is None else b.astype(np.float32)
    g = t.test_math(fn, a, b)
    c = orc.math_eval(fn, orc.MATH_DET, a, b)
    print("math fn", fn, "bit mismatches", int((g.view(np.uint32) != c.view(np.uint32)).sum()))

for which, cam_kw, depth in ((0, pt.CORNELL_CAMERA, 4), (1, pt.DEFAULT_CAMERA, 4)):
    env = pt.synthetic_sky(256, 128) if which == 1 else None
    data = pt.builtin_scene(which, env)
    sc = pt.Scene(t, data)
    nodes, refs = sc.bvh()
    o = orc.Scene(data)
    o.set_flat_bvh(nodes, refs, sc.bvh_width)
    print("scene", which, "nodes", len(nodes), "refs", len(refs), "check", o.check_flat_bvh())
    N = 200000
    org = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    d = rng.normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0 = time.time()
    hg = t.trace_rays(sc, org, d)
    t1 = time.time()
    hc, st = o.trace_rays(orc.ISECT_FLAT, org, d, nthreads=16)
    t2 = time.time()
    bad = 0
    for f in ("t", "u", "v"):
        bad += int((hg[f].view(np.uint32) != hc[f].view(np.uint32)).sum())
    for f in ("tri", "mesh", "front"):
        bad += int((hg[f] != hc[f]).sum())
    print("trace_rays mismatching fields:", bad, "hits", int((hc["tri"] >= 0).sum()), f"gpu {t1 - t0:.3f}s cpu {t2 - t1:.3f}s")

    cam = pt.make_camera(**cam_kw)
    W, H = 256, 256
    s = pt.make_settings(cam, W, H, depth)
    acc = t.malloc(W * H * 16)
    ldr = t.malloc(W * H * 4)
    t.memset(acc, 0, W * H * 16)
    t.set_option("count_pixel_rays", 1)
    t.reset_counters()
    t.render(sc, s, 0, 1, acc, ldr)
    t.synchronize()
    g_rgb, g_rays = t.last_sample(W * H, rays=True)
    cnt = t.counters()
    r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 0, 1, nthreads=16)
    c_rgb = r["sample"].reshape(-1, 3)
    c_rays = r["rays"].reshape(-1)
    print("frame: gpu rays", cnt.rays, "oracle rays", r["stats"].rays, "per-pixel ray count mismatches", int((g_rays != c_rays).sum()))
    err = np.abs(g_rgb - c_rgb)
    rel = err / np.maximum(np.abs(c_rgb), 1e-3)
    print("sample max abs err", float(err.max()), "max rel err", float(rel.max()), "mean", float(c_rgb.mean()), float(g_rgb.mean()))
    g_acc = t.download(acc, (H, W, 4), np.float32)
    g_ldr = t.download(ldr, (H, W, 4), np.uint8)
    print("acc max abs err", float(np.abs(g_acc - r["acc"]).max()), "ldr max diff", int(np.abs(g_ldr.astype(int) - r["ldr"].astype(int)).max()),
          "ldr mismatching bytes", int((g_ldr != r["ldr"]).sum()))
    t.free(acc)
    t.free(ldr)
    t.set_option("count_pixel_rays", 0)

# --- quick throughput look: Cornell 1080p depth 8
data = pt.builtin_scene(0)
sc = pt.Scene(t, data)
cam = pt.make_camera(**pt.CORNELL_CAMERA)
W, H = 1920, 1080
s = pt.make_settings(cam, W, H, 8)
acc = t.malloc(W * H * 16)
t.memset(acc, 0, W * H * 16)
t.render(sc, s, 0, 1, acc, None)
t.synchronize()
t.reset_counters()
t0 = time.time()
t.render(sc, s, 1, 4, acc, None)
t.synchronize()
dt = time.time() - t0
cnt = t.counters()
print(f"cornell 1080p d8: {cnt.rays / 4 / 1e6:.1f} Mrays/frame, {dt / 4 * 1e3:.1f} ms/frame wall, kernel_ms {cnt.kernel_ms / 4:.1f}, {cnt.rays / dt / 1e6:.0f} Mrays/s")
print("DONE")
