"""Recovers the camera of the reference's Default-scene screenshot (tests/golden/ref_default_screenshot.npz, made by
make_ref_screenshot.py from imgs/tracer01_default.png) and writes it to ref_default_camera.json.

The author moved the camera before taking the screenshot, so five numbers are unknown: position, azimuth, zenith (the
window has no roll and its fov is fixed at pi/3, RendererWindow.hpp:77).  They are fitted to the silhouettes the screenshot
shows: pixels are classed by colour (teal sphere, gold cube, green glass sphere, ground quad), the same classes are taken
from the oracle's primary hits, and the camera that maximises the overlap is kept.  A first guess comes from a least-squares
fit of the projected sphere centres, apparent sphere radii and cube centre; Nelder-Mead on the summed 1 - IoU refines it.
Five fitted numbers against the outlines of five objects: what the fit cannot absorb -- relative positions, sizes, the
rotation of the glass cube, which mesh carries which material -- is what tests/test_oracle_render.py checks.

Run in the build container: python tests/golden/fit_default_camera.py   (about two minutes on 8 cores)
The helpers (classes_of_screenshot, primary_mesh_map) are imported by the test.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 200, 120
FOV = float(np.float32(np.pi / 3))
GOLD, GLASS, GROUND, GREEN, TEAL = 0, 1, 2, 3, 4          # mesh order of the Default scene (DefaultScene/tracer.cpp:115-119)


def classes_of_screenshot():
    """Boolean masks (120 x 200, top row first) of the pixels whose colour names an object beyond doubt."""
    img = np.load(os.path.join(HERE, "ref_default_screenshot.npz"))["ldr_small"].astype(np.float64)
    R, G, B = img[..., 0], img[..., 1], img[..., 2]
    return img, {
        TEAL: (G > R + 40) & (B > R + 30) & (G - B < 45),
        GOLD: (R > B + 45) & (G > B + 30) & (R > 125),
        GREEN: (G > R + 35) & (G > B + 35),
        GROUND: (img.min(-1) > 125) & (img.max(-1) - img.min(-1) < 40),
    }


def camera_basis(az, zen):          # RendererWindow.hpp:15-37
    fy, d, fx, fz = np.cos(zen), np.sin(zen), np.cos(az), np.sin(az)
    front = np.array([fx * d, fy, fz * d])
    right = np.array([-fz, 0, fx])
    return front, right, np.cross(right, front)


PIXEL = 1.0 / min(W, H) * float(np.arctan(np.float32(FOV) / 2)) * 2          # GenRay, CUDATracerProg.cu:389-397


def primary_rays(p):
    """Rays through the pixel centres of a 200 x 120 frame, displayed order (top row first)."""
    front, right, up = camera_basis(p[3], p[4])
    u = (np.arange(W) + 0.5) * PIXEL - W * PIXEL / 2
    v = (np.arange(H) + 0.5) * PIXEL - H * PIXEL / 2
    D = front[None, None] + right[None, None] * u[None, :, None] + up[None, None] * v[:, None, None]
    D = (D / np.linalg.norm(D, axis=-1, keepdims=True))[::-1]
    O = np.broadcast_to(np.asarray(p[:3], np.float32), D.shape)
    return np.ascontiguousarray(O.reshape(-1, 3), np.float32), np.ascontiguousarray(D.reshape(-1, 3), np.float32)


def primary_mesh_map(orc, scene, tri_mesh, p):
    O, D = primary_rays(p)
    h, _ = scene.trace_rays(orc.ISECT_BRUTE, O, D)
    t = h["tri"].reshape(H, W)
    return np.where(t >= 0, tri_mesh[np.maximum(t, 0)], -1)


def triangle_meshes(data):
    return np.concatenate([np.full(len(m[1]) // 3, k) for k, m in enumerate(data.meshes)])


def overlaps(mm, masks):
    return {k: float(((mm == k) & m).sum() / max(((mm == k) | m).sum(), 1)) for k, m in masks.items()}


def main():
    from scipy.optimize import least_squares, minimize
    import cudapathtracer_amd as pt
    from oracle import orc
    img, masks = classes_of_screenshot()
    data = pt.builtin_scene(1, None)
    scene = orc.Scene(data)
    tri_mesh = triangle_meshes(data)

    def centroid(m):
        ys, xs = np.nonzero(m)
        return xs.mean(), ys.mean(), m.sum()
    ft = {k: centroid(masks[k]) for k in (TEAL, GOLD, GREEN)}
    spheres = {TEAL: (np.array([-1.2, 1.0, -1.2]), 0.8), GREEN: (np.array([-1.5, 0.5, 1.2]), 0.5)}      # DefaultScene/tracer.cpp:84-92
    cube_centre = np.array([0.5, 0.6, 0.0])

    def project(p, X):
        f, r, u = camera_basis(p[3], p[4])
        v = X - p[:3]
        z = v @ f
        if z <= 0.05:
            return None
        return ((v @ r) / z + W * PIXEL / 2) / PIXEL - 0.5, H - 1 - (((v @ u) / z + H * PIXEL / 2) / PIXEL - 0.5), z

    def resid(p):
        out = []
        for k, (C, rad) in spheres.items():
            q = project(p, C)
            if q is None:
                return np.full(8, 1e3)
            rpix = rad / np.sqrt(max(q[2] ** 2 - rad * rad, 1e-6)) / PIXEL
            out += [q[0] - ft[k][0], q[1] - ft[k][1], 2 * (rpix - np.sqrt(ft[k][2] / np.pi))]
        q = project(p, cube_centre)
        if q is None:
            return np.full(8, 1e3)
        return np.array(out + [q[0] - ft[GOLD][0], q[1] - ft[GOLD][1]])

    rng, best = np.random.default_rng(0), None
    for _ in range(100):
        az, zen, dist = rng.uniform(0, 2 * np.pi), rng.uniform(1.7, 2.6), rng.uniform(2.5, 7)
        p0 = np.array([*(np.array([-0.5, 0.5, 0.0]) - camera_basis(az, zen)[0] * dist), az, zen])
        r = least_squares(resid, p0)
        if best is None or r.cost < best[0]:
            best = (r.cost, r.x)
    p = best[1]

    def cost(q):
        if not 0.05 < q[4] < 3.09:
            return 10.0
        return sum(1 - v for v in overlaps(primary_mesh_map(orc, scene, tri_mesh, q), masks).values())
    simplex = p[None] + np.vstack([np.zeros(5), np.diag([0.1, 0.1, 0.1, 0.03, 0.03])])
    p = minimize(cost, p, method="Nelder-Mead", options=dict(xatol=2e-3, fatol=1e-4, maxiter=250, initial_simplex=simplex)).x
    ov = overlaps(primary_mesh_map(orc, scene, tri_mesh, p), masks)
    out = {"pos": [float(x) for x in p[:3]], "azimuth": float(p[3]), "zenith": float(p[4]), "fov": FOV,
           "fitted_overlap": {"teal": ov[TEAL], "gold": ov[GOLD], "green": ov[GREEN], "ground": ov[GROUND]},
           "note": "fitted to the silhouettes of tests/golden/ref_default_screenshot.npz by tests/golden/fit_default_camera.py"}
    with open(os.path.join(HERE, "ref_default_camera.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
