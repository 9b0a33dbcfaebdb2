"""Generates the golden fixtures in this directory from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself ships no golden vectors for this path and cannot be
built here, so these pin the oracle's own output (regression) and give the GPU tests fixed expected values.

Fixtures:
  rays_<scene>.npz       4096 fixed rays (2048 primary camera rays + 2048 incoherent rays from inside the scene) with
                         the brute-force closest hit (tri, mesh, front, t, u, v)
  render_<scene>_<mode>.npz  a 32x32 crop of a 256x256 render: accumulation buffer, LDR bytes, total rays
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cudapathtracer_amd as pt  # noqa: E402  (scene construction only: host code, no GPU)
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CROP = (112, 112, 144, 144)
CONFIGS = {  # name -> (builtin scene, camera, env size, maxDepth, frames)
    "cornell": (0, pt.CORNELL_CAMERA, None, 4, 16),
    "default": (1, pt.DEFAULT_CAMERA, (64, 32), 6, 8),
}


def scene_data(name):
    which, cam, env, depth, frames = CONFIGS[name]
    e = pt.synthetic_sky(env[0], env[1], 0x5EED) if env else None
    return pt.builtin_scene(which, e)


def fixed_rays(name, n=2048):
    which, cam_kw, env, depth, frames = CONFIGS[name]
    cam = pt.make_camera(**cam_kw)
    rng = np.random.default_rng(1234 + which)
    # primary rays through random image-plane points (same construction as GenRay, fov scale atan(fov/2))
    s = np.float32(np.arctan(np.float32(cam_kw["fov"]) / 2))
    uv = rng.uniform(-s, s, (n, 2)).astype(np.float32)
    f, r, u = (np.array(x[:], np.float32) for x in (cam.front, cam.right, cam.up))
    d1 = f[None] + r[None] * uv[:, :1] + u[None] * uv[:, 1:]
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
    o1 = np.repeat(np.array(cam.pos[:], np.float32)[None], n, 0)
    # incoherent rays from inside the scene's bounds (above every floor: no coplanar ties)
    lo, hi = ((-1.9, -1.9, -1.9), (1.9, 1.9, 1.9)) if which == 0 else ((-3, 0.05, -3), (3, 2.5, 3))
    o2 = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    return np.concatenate([o1, o2]), np.concatenate([d1, d2])


def main():
    for name, (which, cam_kw, env, depth, frames) in CONFIGS.items():
        data = scene_data(name)
        o = orc.Scene(data)
        org, d = fixed_rays(name)
        h, _ = o.trace_rays(orc.ISECT_BRUTE, org, d)
        np.savez_compressed(os.path.join(HERE, f"rays_{name}.npz"), origins=org, dirs=d, hits=h)
        cam = pt.make_camera(**cam_kw)
        s = pt.make_settings(cam, 256, 256, depth)
        for mode, (isect, rng, mth) in {"dfs_libm": (orc.ISECT_REF, orc.RNG_DFS, orc.MATH_LIBM),
                                         "node_det": (orc.ISECT_BRUTE, orc.RNG_NODE, orc.MATH_DET)}.items():
            r = o.render(s, isect, rng, mth, 0, frames, rect=CROP)
            x0, y0, x1, y1 = CROP
            np.savez_compressed(os.path.join(HERE, f"render_{name}_{mode}.npz"), acc=r["acc"][y0:y1, x0:x1],
                                ldr=r["ldr"][y0:y1, x0:x1], rays=np.uint64(r["stats"].rays),
                                last_rays=r["rays"][y0:y1, x0:x1])
            print(name, mode, "rays", r["stats"].rays, "mean", r["acc"][y0:y1, x0:x1, :3].mean())


if __name__ == "__main__":
    main()
