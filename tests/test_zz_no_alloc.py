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


def hip_runtime():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    return hip


def free_bytes(hip):
    import ctypes as C
    f, t = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def test_trace_static_allocates_nothing(tracer, ptlib):
    """free device memory is the same before and after a series of calls of different sizes (the entry point exists for
    callers that cannot allocate); mipt_trace on a fresh size, for contrast, does allocate"""
    hip = hip_runtime()
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    cam = ptlib.make_camera(**ptlib.CORNELL_CAMERA)
    acc, ldr = tracer.malloc(65536 * 16), tracer.malloc(65536 * 4)
    tracer.memset(acc, 0, 65536 * 16)
    tracer.trace_static(sc, ptlib.make_settings(cam, 32, 32, 2), acc, ldr)
    tracer.synchronize()
    before = free_bytes(hip)
    for k, (w, h, d) in enumerate([(256, 256, 8), (33, 17, 5), (128, 64, 8), (1, 1, 1), (256, 255, 7)] * 3):
        s = ptlib.make_settings(cam, w, h, d, frame=k)
        tracer.trace_static(sc, s, acc, ldr)
    tracer.synchronize()
    assert free_bytes(hip) == before
    tracer.free(acc), tracer.free(ldr)


def test_trace_static_under_stream_capture(tracer, ptlib):
    """the use the entry point is for: captured into a HIP graph on the caller's stream (no allocation, no event creation,
    no dependency on the tracer's internal streams), the graph replayed twice; the image equals mipt_trace's"""
    import ctypes as C
    hip = hip_runtime()
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    W, H = 128, 96
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), W, H, 6)
    ref_acc = tracer.malloc(W * H * 16)
    tracer.memset(ref_acc, 0, W * H * 16)
    tracer.render(sc, s, 0, 3, ref_acc)                     # leaves pipelined frames behind: set_stream must join them
    tracer.memset(ref_acc, 0, W * H * 16)
    tracer.reset_counters()
    tracer.trace(sc, s, ref_acc, None)
    ref = tracer.download(ref_acc, (H, W, 4), np.float32)
    rays_one = tracer.counters().rays
    acc, ldr = tracer.malloc(W * H * 16), tracer.malloc(W * H * 4)
    tracer.memset(acc, 0, W * H * 16)
    tracer.render(sc, s, 5, 2, ref_acc)                     # in flight again when the stream is handed over
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    graph, gexec = C.c_void_p(), C.c_void_p()
    try:
        tracer.set_stream(stream.value)
        tracer.reset_counters()
        assert hip.hipStreamBeginCapture(stream, 2) == 0          # hipStreamCaptureModeRelaxed
        try:
            tracer.trace_static(sc, s, acc, ldr)
        finally:
            rc = hip.hipStreamEndCapture(stream, C.byref(graph))
        assert rc == 0 and graph.value
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for _ in range(2):                                          # frame 0 twice: weight 1 overwrites, same image
            assert hip.hipGraphLaunch(gexec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        got = tracer.download(acc, (H, W, 4), np.float32)
        assert tracer.counters().rays == 2 * rays_one               # nothing ran at capture time, two replays did
        assert rel_err(got, ref).max() < RTOL and (got[..., 3] == 1).all()
    finally:
        tracer.set_stream(None)
        if gexec.value:
            hip.hipGraphExecDestroy(gexec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
        tracer.free(acc), tracer.free(ldr), tracer.free(ref_acc)
