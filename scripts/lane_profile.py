"""Lane utilisation per phase of k_wavestack, from the instrumented build (make -C cudapathtracer_amd/csrc prof).
usage: MIPT_LIB=cudapathtracer_amd/libmipt_prof.so python scripts/lane_profile.py cornell|default|grid [node_width] [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
assert "prof" in os.environ.get("MIPT_LIB", ""), "set MIPT_LIB to the instrumented library"
import cudapathtracer_amd as pt
wl = sys.argv[1]; nw = int(sys.argv[2]) if len(sys.argv) > 2 else 0; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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
sc = pt.Scene(t, data, node_width=nw)
s = pt.make_settings(cam, W, H, D)
acc = t.malloc(W * H * 16)
t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
t.render(sc, s, 1, frames, acc); t.synchronize()
c = t.counters(); p = t.level_profile()
k = [p.rays[7 + j] for j in range(12)]
rounds, act, nodeIt, nodeSum, leafIt, triSum, fB, fL, tB, tL, rf, rfL = k
print(f"{wl} width {sc.bvh_width}: {c.kernel_ms / frames:.2f} ms/frame, {c.rays / frames / 1e6:.1f} Mrays/frame (instrumented build)")
print(f"  traversal rounds/ray {rounds * 64 / c.rays:.2f} (x64 lanes); lanes active at round start {act / rounds / 64:.3f}")
print(f"  node loop : {nodeSum / c.rays:.2f} visits/ray, wave iterations {nodeIt}, lane utilisation {nodeSum / (nodeIt * 64):.3f}")
print(f"  leaf loop : {triSum / c.rays:.2f} tests/ray, wave iterations {leafIt}, lane utilisation {triSum / (leafIt * 64):.3f}")
print(f"  shade full: {fB} batches, {fL / max(fB, 1):.1f} lanes/batch;  terminal: {tB} batches, {tL / max(tB, 1):.1f} lanes/batch")
print(f"  refills   : {rf} ({rfL / max(rf, 1):.1f} lanes each)")
# rough VALU-cycle weights per wave iteration (static instruction counts): node visit 55 (BVH2) / 130 (Q4), triangle 67, full shade 620, terminal 230
wn = 130 if sc.bvh_width == 4 else 55
tot = nodeIt * wn + leafIt * 67 + fB * 620 + tB * 230
use = (nodeSum * wn + triSum * 67 + fL * 620 + tL * 230) / 64
print(f"  weighted: node {nodeIt * wn / tot:.2f} leaf {leafIt * 67 / tot:.2f} shade {fB * 620 / tot:.2f} terminal {tB * 230 / tot:.2f} of issue; overall lane utilisation {use / tot:.3f}")
