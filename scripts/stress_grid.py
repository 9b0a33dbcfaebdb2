"""BASELINE config C5 stand-in: displaced-grid stress scene (n x n quads = 2 n^2 triangles), single diffuse material,
synthetic HDRI, 3840x2160, maxDepth 4.  usage: stress_grid.py n [frames]   (node visits / triangle tests per ray for the roofline: bench.py --workload grid)"""
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
