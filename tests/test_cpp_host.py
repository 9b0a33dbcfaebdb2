"""The header-only C++ classes with the reference's names (cudapathtracer_amd/cpp) and the headless apps built on
them: compile check on CPU, end-to-end run on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

APPS = os.path.join(ROOT, "cudapathtracer_amd", "apps")

CALLER = r"""
// a caller written against the REFERENCE's API surface (CornellBox/tracer.cpp style); must compile unchanged
#include "RendererWindow.hpp"
using namespace CUDATracer;
Scene SetupScene() {
    SurfaceMaterial diffuseMat{};
        diffuseMat.color = 1.0f; diffuseMat.roughness = 0.5f; diffuseMat.metallic = 0.0f; diffuseMat.opacity = 1.0f; diffuseMat.n = 2.5f;
    auto floor = ConstructMesh(MakeQuad(), diffuseMat, Math::MatMul(Math::MakeTranslate(Math::vec3f{-2,-2,-2}), Math::MakeScale(Math::vec3f{4,1,4})));
    auto box1 = ConstructMesh(MakeCube(), diffuseMat, Math::MatMul(Math::MakeTranslate(Math::vec3f{ 0.5 - 0.5,-2,-0.5 - 0.5 }),
        Math::MatMul(Math::MakeRotation(Math::Quaternion3f::rotate({ 0,1,0 }, 0.4)), Math::MakeScale(Math::vec3f{ 1,2.5,1 }))));
    auto ball = ConstructMesh(MakeSphere(), diffuseMat, Math::MakeTranslate(Math::vec3f(1.3, 0, 1.7)));
    Scene scn{ {{0.98, 0.97,0.99}, -Math::normalize(Math::vec3f{1,1,1}), 0}, {} };
    scn.objects.push_back(std::move(floor)); scn.objects.push_back(std::move(box1)); scn.objects.push_back(std::move(ball));
    return scn;
}
class AppWindow : public RendererWindow {
public:
    AppWindow() : RendererWindow(SetupScene()) {
        auto& settings = renderSettings->GetMutable<0>();
        settings.maxDepth = 4;
        camParam.pos = { -6,0,0 }; camParam.fov = M_PI / 3; camParam.zenith = M_PI / 2;
        UpdateCam(settings.cam);
    }
};
int main() { TypedBuffer<PathTraceSettings> s; (void)sizeof(s); PathTracer* p = nullptr; (void)p; return 0; }
"""


def test_reference_style_caller_compiles(tmp_path):
    src = tmp_path / "caller.cpp"
    src.write_text(CALLER)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "cudapathtracer_amd", "cpp"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_apps_are_built_and_link_only_libmipt():
    for app in ("cornellbox", "defaultscene"):
        path = os.path.join(APPS, app)
        assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
        ldd = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        assert "libmipt.so" in ldd and "liboracle" not in ldd


@pytest.mark.gpu
def test_cornellbox_app_matches_c_abi_path(tmp_path, tracer, ptlib):
    ppm, pfm = str(tmp_path / "c.ppm"), str(tmp_path / "c.pfm")
    r = subprocess.run([os.path.join(APPS, "cornellbox"), "160", "96", "5", ppm, pfm], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "5 frames" in r.stdout
    raw = open(pfm, "rb").read()
    hdr = b"PF\n160 96\n-1.0\n"
    assert raw.startswith(hdr)
    acc_app = np.frombuffer(raw[len(hdr):], np.float32).reshape(96, 160, 3)
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), 160, 96, 4)
    acc = tracer.malloc(160 * 96 * 16)
    tracer.memset(acc, 0, 160 * 96 * 16)
    ldr = tracer.malloc(160 * 96 * 4)
    tracer.render(sc, s, 0, 5, acc, ldr)
    a = tracer.download(acc, (96, 160, 4), np.float32)
    l = tracer.download(ldr, (96, 160, 4), np.uint8)
    tracer.free(acc), tracer.free(ldr)
    assert np.allclose(acc_app, a[..., :3], rtol=1e-4, atol=1e-6)
    body = open(ppm, "rb").read()
    p6 = b"P6 160 96 255\n"
    img = np.frombuffer(body[len(p6):], np.uint8).reshape(96, 160, 3)
    assert np.abs(img.astype(int) - l[::-1, :, :3].astype(int)).max() <= 1      # PPM is top-down


@pytest.mark.gpu
def test_defaultscene_app_runs(tmp_path):
    ppm = str(tmp_path / "d.ppm")
    r = subprocess.run([os.path.join(APPS, "defaultscene"), "128", "72", "3", ppm], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.frombuffer(open(ppm, "rb").read()[len(b"P6 128 72 255\n"):], np.uint8)
    assert img.max() <= 186 and img.mean() > 5
