"""Where a frame's time goes across waves (instrumented build): when the first / last wave found the pixel counter dry,
when the first / last wave ended, mean wave lifetime.  usage: MIPT_LIB=.../libmipt_prof.so tail_profile.py cornell|default [W H]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
assert "prof" in os.environ.get("MIPT_LIB", "")
import cudapathtracer_amd as pt
wl = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920; H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
if wl == "cornell":
    data, cam, D = pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 8
else:
    data, cam, D = pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)), pt.make_camera(**pt.DEFAULT_CAMERA), 12
t = pt.Tracer(0); sc = pt.Scene(t, data); s = pt.make_settings(cam, W, H, D)
acc = t.malloc(W * H * 16)
t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
t.render(sc, s, 1, 1, acc); t.synchronize()
c = t.counters(); p = t.level_profile()
M = (1 << 64) - 1
t0 = M - p.rays[19]; dry0 = M - p.rays[20]; dry1 = p.rays[21]; end0 = M - p.rays[22]; end1 = p.rays[23]; life = p.rays[24]
waves = 4096
us = lambda x: (x - t0) / 100.0
print(f"{wl} {W}x{H}: kernel {c.kernel_ms:.2f} ms (instrumented); first wave dry at {us(dry0):.0f} us, last wave dry at {us(dry1):.0f} us, "
      f"first wave ends {us(end0):.0f} us, last wave ends {us(end1):.0f} us, mean wave lifetime {life / waves / 100.0:.0f} us")
