"""Where a wave's time goes, phase by phase: the s_memtime clock of the instrumented variant
(make -C cudapathtracer_amd/csrc variant NAME=phasetime EXTRA="-DMIPT_LANE_PROFILE -DMIPT_EXP_PHASETIME").
A wave's ticks between two marks go to the phase that ends at the second mark; waves of one SIMD interleave, so a phase's share is
its share of the waves' residence (issue and waiting), not of the issue slots alone.
usage: MIPT_LIB=cudapathtracer_amd/libmipt_phasetime.so python scripts/exp_phase_time.py cornell|default|grid [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
assert "phasetime" in os.environ.get("MIPT_LIB", ""), "set MIPT_LIB to the phase-clock variant"
import cudapathtracer_amd as pt
wl = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if wl == "cornell":
    data, cam, W, H, D = pt.builtin_scene(0), pt.make_camera(**pt.CORNELL_CAMERA), 1920, 1080, 8
elif wl == "default":
    data, cam, W, H, D = pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED)), pt.make_camera(**pt.DEFAULT_CAMERA), 1920, 1080, 12
else:
    v, i = pt.grid_mesh(2236, 8.0, 1.0, 1)
    sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
    data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8)))], sky, pt.synthetic_sky(2048, 1024, 0x5EED))
    cam, W, H, D = pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3))), 3840, 2160, 4
t = pt.Tracer(0)
sc = pt.Scene(t, data)
s = pt.make_settings(cam, W, H, D)
acc = t.malloc(W * H * 16)
t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
t.render(sc, s, 1, frames, acc); t.synchronize()
c = t.counters(); p = t.level_profile()
k = [p.rays[7 + j] for j in range(12)]
fB, tB, rf = k[6], k[8], k[10]
ticks = k[:6] + [k[9], k[7]]
tot = float(sum(ticks))
names = ["full shade batches", "terminal batches", "end-of-frame check", "node loop", "leaf loop + end-of-leaf rule", "parking",
         "loop top (batch decisions)", "refill"]
print(f"{wl} width {sc.bvh_width}: {c.kernel_ms / frames:.2f} ms/frame (instrumented), {c.rays / frames / 1e6:.1f} Mrays/frame")
for n, x in zip(names, ticks):
    print(f"  {n:30s} {x / tot:6.3f}")
print(f"  per full shade batch {ticks[0] / max(fB, 1):.0f} ticks ({fB} batches), per terminal batch {ticks[1] / max(tB, 1):.0f} ({tB}), per refill {ticks[7] / max(rf, 1):.0f} ({rf})")
