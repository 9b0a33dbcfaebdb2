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


# a caller that touches every member the reference declares on CUDAScene (CUDATracer.hpp:205-226) and the three
# CUDABuffer members no reference app calls (CUDABuffer.hpp:18,20,27); it doubles as the GPU end-to-end test below
ACCESSOR_CALLER = r"""
#include <cstdio>
#include <vector>
#include "RendererWindow.hpp"
using namespace CUDATracer;
typedef struct ihipStream_t* hipStream_t;          // what <hip/hip_runtime.h> declares; this caller needs no HIP headers
int main() {
    SurfaceMaterial mat{}; mat.color = 0.8f; mat.roughness = 0.5f; mat.opacity = 1.0f; mat.n = 2.5f;
    Scene scn{ {{0.98f, 0.97f, 0.99f}, -Math::normalize(Math::vec3f{1, 1, 1}), 5.f}, {} };
    scn.objects.push_back(ConstructMesh(MakeQuad(), mat, Math::MatMul(Math::MakeTranslate(Math::vec3f{-4, 0, -4}), Math::MakeScale(Math::vec3f{8, 1, 8}))));
    scn.objects.push_back(ConstructMesh(MakeCube(), mat, Math::MakeTranslate(Math::vec3f{0.5f, 0.f, -0.5f})));
    CUDAScene cs(scn);
    const SceneInfoHeader& info = cs.GetInfo();
    const MeshInfoHeader& cube = cs.GetMesh(1);
    std::printf("meshes %u verts %u idx %u total %u sky %.2f\n", info.meshCnt, cube.vertexCnt, cube.indicesCnt, cube.TotalSize(), info.skyLight.intensity);
    if (info[1] != &cube || cube.GetVertices()[0].position.x != scn.objects[1].GetVertices()[0].position.x || cube.GetIndices()[2] != scn.objects[1].GetIndices()[2]) return 2;

    const unsigned w = 96, h = 64;
    TypedBuffer<PathTraceSettings> settings;
    auto& st = settings.GetMutable<0>();
    st.viewportWidth = w; st.viewportHeight = h; st.maxDepth = 3; st.frameID = 0;
    Scenes::CameraParams cp; cp.pos = { -3, 1.5f, 0 }; cp.azimuth = 0; cp.zenith = 1.9f; cp.fov = 1.0f;
    Scenes::UpdateCam(cp, st.cam);
    CUDABuffer acc(w * h * 16), ldr(w * h * 4);
    PathTracer* tracer = MakeCUDATracerProg();
    auto mean = [&](int channel) {
        const float* a = (const float*)acc.cpu_data(); double m = 0;
        for (unsigned i = 0; i < w * h; i++) m += a[i * 4 + channel];
        return m / (w * h);
    };
    tracer->Trace(cs, settings, (float*)acc.mutable_gpu_data(), (char*)ldr.mutable_gpu_data());
    const double r0 = mean(0), b0 = mean(2);

    // edit the cube's material and lift it through the mutable accessors; the next Trace sees both
    MeshInfoHeader& mc = cs.GetMeshMutable(1);
    mc.material.color = Math::vec3f{ 0.9f, 0.05f, 0.05f };
    for (unsigned i = 0; i < mc.vertexCnt; i++) mc.GetVertices()[i].position.y += 0.25f;
    cs.GetInfoMutable().skyLight.intensity = 5.f;
    if (!cs.Dirty()) return 3;
    tracer->Trace(cs, settings, (float*)acc.mutable_gpu_data(), (char*)ldr.mutable_gpu_data());
    if (cs.Dirty()) return 4;
    const double r1 = mean(0), b1 = mean(2);
    std::printf("before r %.4f b %.4f  after r %.4f b %.4f\n", r0, b0, r1, b1);

    // the same edit made on the host scene and uploaded as a new CUDAScene must give the same frame
    Scene scn2 = scn;
    {
        std::vector<Vertex> v = scn.objects[1].GetVertices();
        for (auto& x : v) x.position.y += 0.25f;
        SurfaceMaterial red = mat; red.color = Math::vec3f{ 0.9f, 0.05f, 0.05f };
        scn2.objects[1] = Mesh(v, scn.objects[1].GetIndices(), red);
    }
    CUDAScene cs2(scn2);
    CUDABuffer acc2(w * h * 16);
    tracer->Trace(cs2, settings, (float*)acc2.mutable_gpu_data(), (char*)ldr.mutable_gpu_data());
    const float* a = (const float*)acc.cpu_data(); const float* b = (const float*)acc2.cpu_data();
    double worst = 0;
    for (unsigned i = 0; i < w * h * 4; i++) { double e = std::fabs((double)a[i] - b[i]) / std::fmax(std::fabs((double)b[i]), 1e-3); if (e > worst) worst = e; }
    std::printf("refit vs fresh scene: worst relative difference %.3g\n", worst);

    // set_cpu_data / async_gpu_push(stream) / set_gpu_data
    std::vector<float> mine(1024, 1.5f);
    CUDABuffer buf(mine.size() * sizeof(float));
    buf.set_cpu_data(mine.data());
    if (buf.head() != CUDABuffer::HEAD_AT_CPU || buf.cpu_data() != mine.data()) return 5;
    hipStream_t stream = nullptr;                        // the tracer's own stream
    buf.async_gpu_push(stream);
    if (buf.head() != CUDABuffer::SYNCED) return 6;
    CUDABuffer view(mine.size() * sizeof(float));
    view.set_gpu_data(const_cast<void*>(buf.gpu_data()));     // non-owning: `buf` keeps the allocation
    if (view.head() != CUDABuffer::HEAD_AT_GPU) return 7;
    const float* back = (const float*)view.cpu_data();        // blocking D2H after the push on the same stream
    for (size_t i = 0; i < mine.size(); i++) if (back[i] != 1.5f) return 8;
    std::printf("buffers ok\n");
    delete tracer;
    return ((r1 - b1) - (r0 - b0) > 0.005 && worst < 1e-4) ? 0 : 9;     // the frame turned redder; refit == fresh upload
}
"""


def test_reference_style_caller_compiles(tmp_path):
    for name, text in (("caller.cpp", CALLER), ("accessors.cpp", ACCESSOR_CALLER)):
        src = tmp_path / name
        src.write_text(text)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "cudapathtracer_amd", "cpp"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
def test_scene_accessors_and_buffer_setters_end_to_end(tmp_path):
    """CUDAScene::GetMesh / GetMeshMutable / GetInfo / GetInfoMutable and CUDABuffer::set_cpu_data / set_gpu_data /
    async_gpu_push(stream) used the way a reference-style caller would: an edit through the mutable accessors reaches
    the GPU with the next Trace and equals a freshly uploaded scene."""
    libdir = os.path.join(ROOT, "cudapathtracer_amd")
    src = tmp_path / "accessors.cpp"
    src.write_text(ACCESSOR_CALLER)
    exe = str(tmp_path / "accessors")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(libdir, "cpp"), str(src), "-o", exe, "-L", libdir, "-lmipt",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "meshes 2 verts 24 idx 36" in r.stdout and "buffers ok" in r.stdout


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


OBJ_TEXT = """# a unit cube as quads with shared corners, no normals, plus one triangle with explicit normals / relative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0.5 0.25
vn 0 0 1
f 1 4 3 2
f 5 6 7 8
f 1 2 6 5
f 2 3 7 6
f 3 4 8 7
f 4 1 5 8
v 2 0 0
v 3 0 0
v 2 1 0
f -3/1/1 -2/1/1 -1//-1
"""

OBJ_MAIN = r"""
#include <cstdio>
#include "ObjLoader.hpp"
using namespace CUDATracer;
int main(int argc, char** argv) {
    try {
        auto m = LoadOBJ(argv[1]);
        auto& v = std::get<0>(m); auto& i = std::get<1>(m);
        std::printf("%zu %zu\n", v.size(), i.size());
        for (auto& x : v) std::printf("%g %g %g  %.6f %.6f %.6f  %g %g\n", x.position.x, x.position.y, x.position.z, x.normal.x, x.normal.y, x.normal.z, x.uv.x, x.uv.y);
        for (size_t k = 0; k < i.size(); k += 3) std::printf("t %u %u %u\n", i[k], i[k + 1], i[k + 2]);
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
"""


def _build_obj_tool(tmp_path):
    src = tmp_path / "obj_main.cpp"
    src.write_text(OBJ_MAIN)
    exe = str(tmp_path / "obj_main")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "cudapathtracer_amd", "cpp"), str(src), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_obj_loader_parses_polygons_relative_indices_and_smooth_normals(tmp_path):
    exe = _build_obj_tool(tmp_path)
    obj = tmp_path / "cube.obj"
    obj.write_text(OBJ_TEXT)
    r = subprocess.run([exe, str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    nv, ni = map(int, lines[0].split())
    assert (nv, ni) == (8 + 3, 12 * 3 + 3)                       # corners share a vertex per position; quads -> 2 triangles
    verts = np.array([[float(x) for x in l.split()] for l in lines[1:1 + nv]])
    tris = np.array([[int(x) for x in l.split()[1:]] for l in lines[1 + nv:]])
    assert tris.shape == (13, 3) and tris.max() == nv - 1
    assert {tuple(v) for v in verts[:8, 0:3]} == {(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)}   # first-use order
    cube = verts[:8]
    # smooth normals of a cube's corners point along the diagonal, unit length, outward (faces are wound outward)
    assert np.allclose(np.linalg.norm(cube[:, 3:6], axis=1), 1, atol=1e-5)
    assert np.allclose(cube[:, 3:6], (cube[:, 0:3] * 2 - 1) / np.sqrt(3), atol=1e-5)
    # face winding survives fan triangulation: every cube triangle faces away from the centre
    p = cube[:, 0:3]
    for t in tris[:12]:
        n = np.cross(p[t[1]] - p[t[0]], p[t[2]] - p[t[0]])
        assert np.dot(n, p[t].mean(0) - 0.5) > 0
    # the explicit triangle: relative indices, uv and normal taken from the file
    assert np.allclose(verts[8:, 0:3], [(2, 0, 0), (3, 0, 0), (2, 1, 0)]) and np.allclose(verts[8:, 3:6], (0, 0, 1))
    assert np.allclose(verts[8:10, 6:8], (0.5, 0.25)) and np.allclose(verts[10, 6:8], (0, 0))


@pytest.mark.parametrize("bad,msg", [("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "out of range"), ("v 0 0 0\nv 1 0 0\nf 1 2\n", "fewer than 3"),
                                     ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 x 3\n", "bad face corner"), ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "index 0")])
def test_obj_loader_rejects_malformed_faces(tmp_path, bad, msg):
    exe = _build_obj_tool(tmp_path)
    obj = tmp_path / "bad.obj"
    obj.write_text(bad)
    r = subprocess.run([exe, str(obj)], capture_output=True, text=True)
    assert r.returncode == 1 and msg in r.stdout and "OBJ line" in r.stdout
    r = subprocess.run([exe, str(tmp_path / "missing.obj")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stdout


WINDOW_MAIN = r"""
// drives the headless RendererWindow the way GLFW would: frames, key state, mouse drag, resize, preview files
#include <cstdio>
#include "RendererWindow.hpp"
using namespace CUDATracer;
struct Wnd : RendererWindow {
    using RendererWindow::RendererWindow;
    const PathTraceSettings& st() { return renderSettings->Get<0>(); }
};
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main(int argc, char** argv) {
    try {
        Wnd w(Scenes::CornellBox(), "t", 64, 40);
        w.camParam = Scenes::CornellCamera();
        w.SetPreview(argv[1], 2);
        w.run(5);
        EXPECT(w.frameID() == 5);
        EXPECT(w.frame().size() == 64 * 40 * 4);
        Math::vec3f p0 = w.camParam.pos;
        w.keys.w = true; w.Update(); w.keys.w = false;                 // W: 0.1 along front, accumulation restarts
        EXPECT(w.frameID() == 0);
        EXPECT(std::fabs(w.camParam.pos.x - (p0.x + 0.1f * w.st().cam.front.x)) < 1e-6f);
        w.render(); w.render();
        EXPECT(w.frameID() == 2);
        w.Update();                                                     // no key: nothing changes
        EXPECT(w.frameID() == 2);
        w.mouseMotion(10, 10);                                          // button up: motion is tracked, camera is not touched
        EXPECT(w.frameID() == 2);
        float az = w.camParam.azimuth, ze = w.camParam.zenith;
        w.mouseLeftDown = true; w.mouseMotion(50, 30);
        EXPECT(w.frameID() == 0);
        EXPECT(std::fabs(w.camParam.azimuth - (az + 40.f / 400)) < 1e-6f && std::fabs(w.camParam.zenith - (ze + 20.f / 400)) < 1e-6f);
        w.mouseMotion(50, 30);                                          // no movement: no reset
        w.render();
        EXPECT(w.frameID() == 1);
        w.resize({ 32, 16 });                                           // new buffers, accumulation restarts
        EXPECT(w.frameID() == 0 && w.st().viewportWidth == 32 && w.st().viewportHeight == 16);
        w.render();
        EXPECT(w.frame().size() == 32 * 16 * 4 && w.frameID() == 1);
        const float* a = w.accumulation();
        EXPECT(a[3] == 1.0f);                                           // alpha written by the kernel
        // animation step: the light goes out -> UpdateScene re-fits in place, accumulation restarts, the image is black
        Scene lit = Scenes::CornellBox(), dark = lit;
        dark.objects.clear();
        for (const Mesh& m : lit.objects) { SurfaceMaterial mat = m.GetMat(); mat.emissiveStrength = 0.f; dark.objects.emplace_back(m.GetVertices(), m.GetIndices(), mat); }
        w.UpdateScene(dark);
        EXPECT(w.frameID() == 0);
        w.render();
        a = w.accumulation();
        float sum = 0.f; for (int i = 0; i < 32 * 16; i++) sum += a[4 * i] + a[4 * i + 1] + a[4 * i + 2];
        EXPECT(sum == 0.f);
        w.UpdateScene(Scenes::CornellBox());
        w.render();
        a = w.accumulation();
        sum = 0.f; for (int i = 0; i < 32 * 16; i++) sum += a[4 * i] + a[4 * i + 1] + a[4 * i + 2];
        EXPECT(sum > 0.f);
        bool threw = false;
        try { w.UpdateScene(Scenes::DefaultScene()); } catch (const std::exception&) { threw = true; }   // another topology
        EXPECT(threw);
        std::printf("OK\n");
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
"""


@pytest.mark.gpu
def test_renderer_window_input_resets_accumulation_and_preview_files(tmp_path):
    src = tmp_path / "wnd.cpp"
    src.write_text(WINDOW_MAIN)
    exe = str(tmp_path / "wnd")
    libdir = os.path.join(ROOT, "cudapathtracer_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(libdir, "cpp"), str(src), "-o", exe, "-L", libdir, "-lmipt",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    prefix = str(tmp_path / "prev")
    r = subprocess.run([exe, prefix], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    for k in (2, 4):
        body = open(f"{prefix}_{k}.ppm", "rb").read()
        assert body.startswith(b"P6 64 40 255\n") and len(body) == len(b"P6 64 40 255\n") + 64 * 40 * 3
    assert not os.path.exists(f"{prefix}_5.ppm")


CHECKPOINT_MAIN = r"""
// checkpoint / resume of a progressive render: 6 frames in one go against 3 frames, checkpoint, a NEW window, 3 more frames
#include <cstdio>
#include <cmath>
#include "RendererWindow.hpp"
using namespace CUDATracer;
struct Wnd : RendererWindow {
    using RendererWindow::RendererWindow;
    const PathTraceSettings& st() { return renderSettings->Get<0>(); }
    void depth(unsigned d) { renderSettings->GetMutable<0>().maxDepth = d; }
};
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main(int argc, char** argv) {
    try {
        const std::string ck = argv[1];
        const int W = 48, H = 32;
        std::vector<float> whole;
        {
            Wnd w(Scenes::CornellBox(), "t", W, H);
            w.camParam = Scenes::CornellCamera(); w.camParam.azimuth += 0.05f;
            w.resize({ W, H }); w.depth(5);
            for (int f = 0; f < 6; f++) { w.Update(); w.render(); }
            whole.assign(w.accumulation(), w.accumulation() + (size_t)W * H * 4);
        }
        {
            Wnd w(Scenes::CornellBox(), "t", W, H);
            w.camParam = Scenes::CornellCamera(); w.camParam.azimuth += 0.05f;
            w.resize({ W, H }); w.depth(5);
            for (int f = 0; f < 3; f++) { w.Update(); w.render(); }
            EXPECT(w.SaveCheckpoint(ck));
        }
        Wnd r(Scenes::CornellBox(), "t", 8, 8);                      // another size, another camera, another depth: all restored
        EXPECT(!r.LoadCheckpoint(ck + ".missing"));
        EXPECT(r.LoadCheckpoint(ck));
        EXPECT(r.frameID() == 3 && r.st().viewportWidth == W && r.st().viewportHeight == H && r.st().maxDepth == 5);
        for (int f = 0; f < 3; f++) { r.Update(); r.render(); }
        EXPECT(r.frameID() == 6);
        const float* a = r.accumulation();
        double num = 0, den = 0; float worst = 0;
        for (size_t i = 0; i < whole.size(); i++) { num += std::fabs(a[i] - whole[i]); den += std::fabs(whole[i]); worst = std::fmax(worst, std::fabs(a[i] - whole[i]) / (1e-3f + std::fabs(whole[i]))); }
        std::printf("resumed vs uninterrupted: mean rel %.3g worst %.3g\n", num / den, worst);
        EXPECT(den > 0 && num / den < 1e-6 && worst < 2e-5f);        // only the order of the adds into a pixel differs
        // a damaged or truncated file is refused and changes nothing
        FILE* f = std::fopen(ck.c_str(), "rb"); std::vector<unsigned char> bytes((size_t)64 + (size_t)W * H * 16); EXPECT(std::fread(bytes.data(), 1, bytes.size(), f) == bytes.size()); std::fclose(f);
        auto put = [&](const std::string& p, size_t n) { FILE* g = std::fopen(p.c_str(), "wb"); std::fwrite(bytes.data(), 1, n, g); std::fclose(g); };
        put(ck + ".short", bytes.size() - 5);
        EXPECT(!r.LoadCheckpoint(ck + ".short") && r.frameID() == 6);
        bytes[0] ^= 1; put(ck + ".magic", bytes.size());
        EXPECT(!r.LoadCheckpoint(ck + ".magic") && r.frameID() == 6);
        std::printf("OK\n");
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
"""


@pytest.mark.gpu
def test_renderer_window_checkpoint_and_resume(tmp_path):
    """SURVEY §5 (checkpoint / resume, optional): accumulation + sample index + camera saved by one window and resumed by
    another give the image of the uninterrupted run (same samples; only the order of the atomic adds differs)."""
    src = tmp_path / "ck.cpp"
    src.write_text(CHECKPOINT_MAIN)
    exe = str(tmp_path / "ck")
    libdir = os.path.join(ROOT, "cudapathtracer_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(libdir, "cpp"), str(src), "-o", exe, "-L", libdir, "-lmipt",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, str(tmp_path / "render.ckpt")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert os.path.getsize(tmp_path / "render.ckpt") == 64 + 48 * 32 * 16 and not os.path.exists(tmp_path / "render.ckpt.tmp")


@pytest.mark.gpu
def test_objscene_app_loads_builds_caches_and_renders(tmp_path):
    # a 64x64 height-field written as OBJ quads (8 k triangles), rendered twice: second run takes the BVH from the cache
    n = 64
    xs = np.linspace(-1.5, 1.5, n + 1)
    with open(tmp_path / "hf.obj", "w") as f:
        for z in xs:
            for x in xs:
                f.write(f"v {x:.6f} {0.3 + 0.2 * np.sin(3 * x) * np.cos(2 * z):.6f} {z:.6f}\n")
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i + 1
                f.write(f"f {a} {a + n + 1} {a + n + 2} {a + 1}\n")
    cache = str(tmp_path / "hf.bvh")
    outs = []
    for run in range(2):
        ppm, pfm = str(tmp_path / f"o{run}.ppm"), str(tmp_path / f"o{run}.pfm")
        r = subprocess.run([os.path.join(APPS, "objscene"), str(tmp_path / "hf.obj"), "96", "64", "4", ppm, pfm, cache],
                           capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{2 * n * n} triangles loaded" in r.stdout
        assert ("from cache" if run else "BVH built") in r.stdout
        outs.append(open(pfm, "rb").read())
    hdr = b"PF\n96 64\n-1.0\n"
    a, b = (np.frombuffer(o[len(hdr):], np.float32) for o in outs)
    assert np.allclose(a, b, rtol=1e-4, atol=1e-6) and a.mean() > 0.01      # same BVH -> same image (up to add order)
    # the same app with the tree built on the GPU ($MIPT_BVH_BUILDER=2 -> CUDAScene): closest hits do not depend on the tree,
    # so neither does the image
    pfm = str(tmp_path / "gpu.pfm")
    r = subprocess.run([os.path.join(APPS, "objscene"), str(tmp_path / "hf.obj"), "96", "64", "4", str(tmp_path / "gpu.ppm"), pfm],
                       capture_output=True, text=True, timeout=180, env=dict(os.environ, MIPT_BVH_BUILDER="2", MIPT_BUILD_VERBOSE="1"))
    assert r.returncode == 0 and "GPU BVH build: 8194 triangles" in r.stderr          # the height field + the app's ground quad, r.stdout + r.stderr
    g = np.frombuffer(open(pfm, "rb").read()[len(hdr):], np.float32)
    assert np.allclose(a, g, rtol=1e-4, atol=1e-6)


PLY_MAIN = r"""
#include <cstdio>
#include "PlyLoader.hpp"
using namespace CUDATracer;
int main(int argc, char** argv) {
    try {
        auto m = LoadPLY(argv[1]);
        auto& v = std::get<0>(m); auto& i = std::get<1>(m);
        std::printf("%zu %zu\n", v.size(), i.size());
        for (auto& x : v) std::printf("%g %g %g  %.6f %.6f %.6f  %g %g\n", x.position.x, x.position.y, x.position.z, x.normal.x, x.normal.y, x.normal.z, x.uv.x, x.uv.y);
        for (size_t k = 0; k < i.size(); k += 3) std::printf("t %u %u %u\n", i[k], i[k + 1], i[k + 2]);
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
"""


def _ply_files(tmp_path):
    """the same pyramid (square base as one quad, four sides) as ascii without normals and as little-endian binary with
    normals, texture coordinates, an extra vertex property, an extra list property on the faces and a trailing element"""
    import struct
    pos = [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1.5, 0)]
    faces = [(0, 1, 2, 3), (0, 4, 1), (1, 4, 2), (2, 4, 3), (3, 4, 0)]
    a = tmp_path / "pyr_ascii.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment made by the test\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 5\nproperty list uchar int vertex_indices\nend_header\n" +
                 "".join(f"{x} {y} {z}\n" for x, y, z in pos) + "".join(f"{len(f)} " + " ".join(map(str, f)) + "\n" for f in faces))
    b = tmp_path / "pyr_bin.ply"
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\n"
           "property uchar red\nproperty float nx\nproperty float ny\nproperty float nz\nproperty float s\nproperty float t\n"
           "element face 5\nproperty list uchar ushort vertex_index\nproperty list uchar float texcoord\n"
           "element edge 1\nproperty int vertex1\nproperty int vertex2\nend_header\n").encode()
    body = b""
    for k, (x, y, z) in enumerate(pos):
        body += struct.pack("<dddBfffff", x, y, z, 200, 0.0, 1.0, 0.0, 0.25 * k, 0.5)
    for f in faces:
        body += struct.pack("<B" + "H" * len(f), len(f), *f) + struct.pack("<Bff", 2, 0.1, 0.2)
    body += struct.pack("<ii", 0, 1)
    b.write_bytes(hdr + body)
    return a, b


def test_ply_loader_ascii_and_binary(tmp_path):
    src = tmp_path / "ply_main.cpp"
    src.write_text(PLY_MAIN)
    exe = str(tmp_path / "ply_main")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "cudapathtracer_amd", "cpp"), str(src), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = _ply_files(tmp_path)
    outs = []
    for f in (a, b):
        r = subprocess.run([exe, str(f)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        lines = r.stdout.strip().splitlines()
        assert lines[0] == "5 18"                                     # 1 quad -> 2 triangles, + 4 sides
        verts = np.array([[float(x) for x in l.split()] for l in lines[1:6]])
        tris = np.array([[int(x) for x in l.split()[1:]] for l in lines[6:]])
        outs.append((verts, tris))
    (va, ta), (vb, tb) = outs
    assert np.array_equal(ta, tb) and np.array_equal(ta[:2], [(0, 1, 2), (0, 2, 3)]) and np.allclose(va[:, :3], vb[:, :3])
    # ascii file: smooth normals, unit length; the apex normal is the (area-weighted) mean of four symmetric sides: +y
    assert np.allclose(np.linalg.norm(va[:, 3:6], axis=1), 1, atol=1e-5) and np.allclose(va[4, 3:6], (0, 1, 0), atol=1e-6)
    # binary file: normals and s/t come from the file, unrelated properties are skipped
    assert np.allclose(vb[:, 3:6], (0, 1, 0)) and np.allclose(vb[:, 6], 0.25 * np.arange(5)) and np.allclose(vb[:, 7], 0.5)
    # malformed inputs
    bad = tmp_path / "bad.ply"
    for text, msg in (("plx\n", "magic"), ("ply\nformat binary_big_endian 1.0\nend_header\n", "unsupported format"),
                      ("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n", "without x y z"),
                      ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                       "property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 3\n", "out of range"),
                      ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n1 0\n", "end of data")):
        bad.write_text(text)
        r = subprocess.run([exe, str(bad)], capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stdout, (text, r.stdout)
