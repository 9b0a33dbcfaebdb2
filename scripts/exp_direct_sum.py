"""drift of the direct sum mode (the integrator adds every ray's contribution straight into the caller's running
sum) against the sample-buffer + fold path, small viewport, many frames.  Prints per scene the relative difference of the
image mean (bias) and the per-pixel differences."""
import sys
import numpy as np
sys.path.insert(0, '.')
import cudapathtracer_amd as pt
t = pt.Tracer(0)
W = H = 32
for name in ("cornell", "default"):
    if name == "cornell":
        sc = pt.Scene(t, pt.builtin_scene(0)); s = pt.make_settings(pt.make_camera(**pt.CORNELL_CAMERA), W, H, 8)
    else:
        sc = pt.Scene(t, pt.builtin_scene(1, pt.synthetic_sky(256, 128, 0x5EED))); s = pt.make_settings(pt.make_camera(**pt.DEFAULT_CAMERA), W, H, 12)
    for N in (256, 2048, 8192):
        acc, ldr = t.malloc(W * H * 16), t.malloc(W * H * 4)
        out = []
        for direct in (True, False):
            t.memset(acc, 0, W * H * 16)
            t.render(sc, s, 0, N, acc, None if direct else ldr, 1, 1)
            t.synchronize()
            out.append(t.download(acc, (H, W, 4), np.float32)[..., :3].astype(np.float64) / N)
        # float64 reference: per-frame samples summed on the host
        d, f = out
        rel = (d - f) / np.maximum(np.abs(f), 1e-3)
        print(f"{name} N={N}: mean(direct)/mean(fold) - 1 = {d.mean() / f.mean() - 1:+.3e}; per pixel rel diff mean {rel.mean():+.3e} max |.| {np.abs(rel).max():.3e}; image mean {f.mean():.4f} max {f.max():.2f}", flush=True)
        t.free(acc); t.free(ldr)
    sc.close()
