"""BASELINE config C5 stand-in: displaced-grid stress scene (n x n quads = 2 n^2 triangles), single diffuse material,
synthetic HDRI, 3840x2160, maxDepth 4.  usage: stress_grid.py n [frames]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cudapathtracer_amd as pt
n = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.time(); v, i = pt.grid_mesh(n, 8.0, 1.0, 1); print(f"mesh {len(i)//3} tris in {time.time()-t0:.1f}s", flush=True)
sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8), roughness=0.5, metallic=0.0, opacity=1.0, n=2.5))], sky, pt.synthetic_sky(2048, 1024))
t = pt.Tracer(0)
t0 = time.time(); sc = pt.Scene(t, data); print(f"scene (SBVH build + upload) in {time.time()-t0:.1f}s", flush=True)
nodes, refs = sc.bvh(); print(f"nodes {len(nodes)} ({len(nodes)*64/1e6:.0f} MB) refs {len(refs)} ({len(refs)*48/1e6:.0f} MB)", flush=True)
W, H, D = 3840, 2160, 4
cam = pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3)))
s = pt.make_settings(cam, W, H, D)
acc = t.malloc(W * H * 16)
t.render(sc, s, 0, 1, acc); t.synchronize(); t.reset_counters()
t0 = time.perf_counter(); t.render(sc, s, 1, frames, acc); t.synchronize(); dt = time.perf_counter() - t0
c = t.counters()
print(f"grid {n}: {dt/frames*1e3:.2f} ms/frame, {c.rays/frames/1e6:.1f} Mrays/frame, {c.rays/dt/1e6:.0f} Mrays/s", flush=True)
if len(sys.argv) > 3:   # oracle sample for N_node / N_tri
    from oracle import orc
    o = orc.Scene(data); o.set_flat_bvh(nodes, refs, sc.bvh_width)
    st_tot = [0, 0, 0]
    for y in range(0, H, H // 8):
        r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 1, 1, rect=(0, y, W, y + 1), want_ldr=False, nthreads=16)
        st_tot[0] += r["stats"].rays; st_tot[1] += r["stats"].node_visits; st_tot[2] += r["stats"].tri_tests
    nn, nt = st_tot[1] / st_tot[0], st_tot[2] / st_tot[0]
    b = 64 * nn + 48 * nt + 64
    print(f"oracle sample: N_node {nn:.2f} N_tri {nt:.2f} B_ray {b:.0f} -> algorithmic {c.rays/dt*b/1e9:.0f} GB/s", flush=True)
