"""The allocation-free entry point mipt_trace_static against the entry point it is a variant of, mipt_trace, through the
C-ABI: same scene, same settings, two calls in a row, the second with a smaller input (the shared static scratch keeps
the first call's contents).  Bar as for every radiance comparison here: identical ray totals, radiance within 2e-5
relative (the order of fp32 atomic adds into a pixel is the only difference), LDR bytes within 1."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu

RTOL, FLOOR = 2e-5, 1e-3


def rel_err(got, exp):
    return np.abs(got - exp) / np.maximum(np.abs(exp), FLOOR)


def run_pair(tracer, ptlib, sc, cam_kw, w, h, depth, frames):
    s = ptlib.make_settings(ptlib.make_camera(**cam_kw), w, h, depth)
    out = []
    for fn in (tracer.trace_static, tracer.trace):
        acc, ldr = tracer.malloc(w * h * 16), tracer.malloc(w * h * 4)
        tracer.memset(acc, 0, w * h * 16)
        tracer.reset_counters()
        for f in frames:
            s.frameID = f
            fn(sc, s, acc, ldr)
        a = tracer.download(acc, (h, w, 4), np.float32)
        l = tracer.download(ldr, (h, w, 4), np.uint8)
        out.append((a, l, tracer.counters().rays))
        tracer.free(acc), tracer.free(ldr)
    (a0, l0, r0), (a1, l1, r1) = out
    assert r0 == r1 and r0 > 0                                   # the same ray trees
    assert rel_err(a0, a1).max() < RTOL and (a0[..., 3] == 1).all()
    assert np.abs(l0.astype(int) - l1.astype(int)).max() <= 1 and (l0[..., 3] == 255).all()
    return a0


def test_trace_static_equals_trace_twice_second_call_smaller(tracer, ptlib):
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    a_big = run_pair(tracer, ptlib, sc, ptlib.CORNELL_CAMERA, 256, 256, 8, (0, 1, 2))       # the limits: 65536 pixels, 8 bounces
    assert a_big[..., :3].max() > 0
    run_pair(tracer, ptlib, sc, ptlib.CORNELL_CAMERA, 67, 33, 3, (0, 4))                   # smaller frame, shallower tree
    sd = ptlib.Scene(tracer, ptlib.builtin_scene(1, ptlib.synthetic_sky(256, 128, 3)))       # 4-wide nodes, env map, refraction
    run_pair(tracer, ptlib, sd, ptlib.DEFAULT_CAMERA, 160, 96, 6, (0, 1))
    run_pair(tracer, ptlib, sd, ptlib.DEFAULT_CAMERA, 1, 1, 1, (0,))


def test_trace_static_rejects_sizes_beyond_the_scratch(tracer, ptlib):
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    cam = ptlib.make_camera(**ptlib.CORNELL_CAMERA)
    acc = tracer.malloc(512 * 256 * 16)
    with pytest.raises(ptlib.MiptError, match="65536 pixels"):
        tracer.trace_static(sc, ptlib.make_settings(cam, 512, 256, 4), acc, None)
    with pytest.raises(ptlib.MiptError, match="maxDepth exceeds 8"):
        tracer.trace_static(sc, ptlib.make_settings(cam, 64, 64, 9), acc, None)
    tracer.set_option("estimator", 1)
    try:
        with pytest.raises(ptlib.MiptError, match="estimator 0 or 2"):
            tracer.trace_static(sc, ptlib.make_settings(cam, 64, 64, 4), acc, None)
    finally:
        tracer.set_option("estimator", 0)
    tracer.free(acc)
