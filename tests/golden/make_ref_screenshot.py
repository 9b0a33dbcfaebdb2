"""Turns the one expected-output artefact the reference holds for the Cornell box -- its README screenshot,
/root/reference/imgs/tracer02_cornellbox.png (README.md:9) -- into a small data fixture (run in the build container,
where /root/reference exists:  python tests/golden/make_ref_screenshot.py).

The PNG is a Windows screenshot of the 800x480 GLFW window (RendererWindow.hpp:60) including chrome: a 3-pixel border
left and right, one border row at the bottom and a 31-row title bar at the top.  The client area [31:511, 3:803] is
box-downsampled 20x20 -> 40 x 24 RGB means (and standard deviations) of the LDR bytes (top row first, as displayed).  spp of the screenshot is
unknown (it is visibly converged); the camera is the app's (CornellBox/tracer.cpp:151-157), maxDepth 4.
Only these 2 x 2 880 numbers are stored -- data, not source.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/imgs/tracer02_cornellbox.png"
CLIENT = (31, 511, 3, 803)          # y0, y1, x0, x1 of the 480 x 800 client area inside the 512 x 806 screenshot


def default_scene():
    """The other screenshot, imgs/tracer01_default.png (README.md:5): the Default scene under the author's HDRI, from a camera the
    author had moved (the app's start camera looks along +x at eye height, this one looks down on the table).  801 x 514 with
    chrome; client area [32:512, 1:801], box-downsampled 4x4 -> 120 x 200 RGB bytes.  Neither the HDRI nor the camera is in the
    reference, so this fixture pins geometry and material hues only (tests/golden/fit_default_camera.py recovers the camera)."""
    a = np.asarray(Image.open("/root/reference/imgs/tracer01_default.png").convert("RGB")).astype(np.float64)
    assert a.shape == (514, 801, 3), a.shape
    border = np.array([24, 131, 215])
    assert (a[32:512, 0] == border).all(-1).mean() > 0.9 and (a[512, 1:800] == border).all(-1).mean() > 0.9
    c = a[32:512, 1:801]
    small = np.rint(c.reshape(120, 4, 200, 4, 3).mean((1, 3))).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_default_screenshot.npz"), ldr_small=small, client=np.array((32, 512, 1, 801)),
                        source=np.array("0x5b25/CUDAPathTracer imgs/tracer01_default.png (README.md:5), client area 800x480 box-downsampled 4x4"))
    print("wrote ref_default_screenshot.npz", small.shape)


def main():
    default_scene()
    a = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
    assert a.shape == (512, 806, 3), a.shape
    y0, y1, x0, x1 = CLIENT
    border = np.array([24, 131, 215])
    assert (a[:, x0 - 1] == border).all(-1).mean() > 0.9 and (a[:, x1] == border).all(-1).mean() > 0.9       # the frame around the client area
    c = a[y0:y1, x0:x1]
    blocks = c.reshape(24, 20, 40, 20, 3)
    small = blocks.mean((1, 3)).astype(np.float32)
    spread = blocks.std((1, 3)).astype(np.float32)          # large inside a block that straddles an edge (wall / opening, light / ceiling)
    np.savez_compressed(os.path.join(HERE, "ref_cornell_screenshot.npz"), ldr_mean=small, ldr_std=spread, client=np.array(CLIENT),
                        source=np.array("0x5b25/CUDAPathTracer imgs/tracer02_cornellbox.png (README.md:9), client area 800x480 box-downsampled 20x20"))
    print("wrote ref_cornell_screenshot.npz", small.shape, float(small.max()))


if __name__ == "__main__":
    main()
