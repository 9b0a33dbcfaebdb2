"""The C++ single-process multi-GPU host (include/mipt_dist.h, libmipt_dist.so = libmipt + RCCL): partition arithmetic
and exports on the CPU, the whole flow with G = 1 on the GPU box (tracers before communicators, strided sum-mode
render, reduce, scale + tonemap on the root) against the single-GPU path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "cudapathtracer_amd", "libmipt_dist.so")


def load():
    assert os.path.exists(LIB), f"{LIB} missing: run __graft_entry__.build()"
    return C.CDLL(LIB)


def test_dist_header_symbols_exported():
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mipt_dist.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(mipt_dist_[a-z0-9_]+)\s*\(", txt)))
    assert len(syms) >= 9
    lib = load()
    assert not [s for s in syms if not hasattr(lib, s)]
    ldd = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libmipt.so" in ldd and "liboracle" not in ldd and "torch" not in ldd


def test_frames_for_rank_partitions_every_frame_exactly_once():
    lib = load()
    lib.mipt_dist_frames_for_rank.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    from cudapathtracer_amd import dist as pydist
    for first, n, world in ((0, 4096, 8), (7, 5, 8), (3, 1, 2), (100, 0, 4), (0, 13, 1), (5, 64, 3)):
        seen = []
        for r in range(world):
            f, c = C.c_uint32(), C.c_uint32()
            assert lib.mipt_dist_frames_for_rank(first, n, r, world, C.byref(f), C.byref(c)) == 0
            assert (f.value, c.value, world) == pydist.frames_for_rank(first, n, r, world)       # same rule as the torch.distributed path
            seen += [f.value + k * world for k in range(c.value)]
        assert sorted(seen) == list(range(first, first + n))
        counts = [len([x for x in seen if (x - first) % world == r]) for r in range(world)]
        assert max(counts) - min(counts) <= 1                                                    # balanced to one frame
    f, c = C.c_uint32(), C.c_uint32()
    assert lib.mipt_dist_frames_for_rank(0, 8, 2, 2, C.byref(f), C.byref(c)) != 0
    assert lib.mipt_dist_frames_for_rank(0, 8, 0, 0, C.byref(f), C.byref(c)) != 0


def test_dist_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = load()
    h = C.c_void_p()
    assert lib.mipt_dist_create(0, C.byref(h)) != 0 and not h.value


@pytest.mark.gpu
def test_multigpu_app_with_one_gpu_equals_the_single_gpu_render(tmp_path, tracer, ptlib):
    exe = os.path.join(ROOT, "cudapathtracer_amd", "apps", "multigpu")
    assert os.path.exists(exe)
    ppm, pfm = str(tmp_path / "m.ppm"), str(tmp_path / "m.pfm")
    w, h, frames, depth = 160, 96, 6, 5
    r = subprocess.run([exe, "cornell", str(w), str(h), str(frames), "1", str(depth), ppm, pfm], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{frames} frames on 1 GPU(s)" in r.stdout
    raw = open(pfm, "rb").read()
    hdr = f"PF\n{w} {h}\n-1.0\n".encode()
    mean = np.frombuffer(raw[len(hdr):], np.float32).reshape(h, w, 3)
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), w, h, depth)
    acc, ldr = tracer.malloc(w * h * 16), tracer.malloc(w * h * 4)
    tracer.memset(acc, 0, w * h * 16)
    tracer.reset_counters()
    tracer.render(sc, s, 0, frames, acc, ldr)
    a = tracer.download(acc, (h, w, 4), np.float32)
    l = tracer.download(ldr, (h, w, 4), np.uint8)
    rays = tracer.counters().rays
    tracer.free(acc), tracer.free(ldr)
    assert f"({rays} rays)" in r.stdout                                   # same ray trees
    assert np.allclose(mean, a[..., :3], rtol=1e-4, atol=1e-6)            # sum / N against the running mean
    img = np.frombuffer(open(ppm, "rb").read()[len(f"P6 {w} {h} 255\n".encode()):], np.uint8).reshape(h, w, 3)
    assert np.abs(img.astype(int) - l[::-1, :, :3].astype(int)).max() <= 1
