// scenes.cpp -- C-ABI access to the callers' scene construction (include/mipt.h, last section), implemented with the
// header-only CUDATracer host API in ../cpp so that Python tests and C++ callers build byte-identical inputs.
#include "../../include/mipt.h"
#include "../cpp/Scenes.hpp"
#include "../cpp/Rgbe.hpp"

#include <algorithm>
#include <cstring>
#include <new>

using namespace CUDATracer;

static_assert(sizeof(Vertex) == sizeof(mipt_vertex) && sizeof(Vertex) == 32, "Vertex layout");
static_assert(sizeof(SurfaceMaterial) == sizeof(mipt_material) && sizeof(SurfaceMaterial) == 48, "SurfaceMaterial layout");
static_assert(sizeof(SkyLight) == sizeof(mipt_skylight) && sizeof(SkyLight) == 28, "SkyLight layout");
static_assert(sizeof(Camera) == sizeof(mipt_camera) && sizeof(Camera) == 52, "Camera layout");
static_assert(sizeof(mipt_settings) == 72, "PathTraceSettings layout");

namespace {
uint32_t copy_out(const std::vector<Vertex>& v, const std::vector<uint32_t>& e, mipt_vertex* vo, uint32_t vcap, uint32_t* io, uint32_t icap, uint32_t* icnt) {
    if (vo) memcpy(vo, v.data(), sizeof(Vertex) * std::min<size_t>(vcap, v.size()));
    if (io) memcpy(io, e.data(), 4 * std::min<size_t>(icap, e.size()));
    if (icnt) *icnt = (uint32_t)e.size();
    return (uint32_t)v.size();
}
const Scene& builtin(int which) {
    static const Scene cornell = Scenes::CornellBox(), def = Scenes::DefaultScene();
    return which == 0 ? cornell : def;
}
}  // namespace

extern "C" {

// No entry point throws (include/mipt.h): host allocations are guarded; the count-returning helpers report 0.
#define MIPT_GUARD_BEGIN try {
#define MIPT_GUARD_END(fallback_) } catch (...) { return fallback_; }

uint32_t mipt_make_mesh(int kind, const float* xform, mipt_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt) {
    MIPT_GUARD_BEGIN
    MeshArrays arr = kind == 0 ? MakeQuad() : kind == 1 ? MakeCube() : MakeSphere();
    Math::mat4x4f T(1.f);
    if (xform) memcpy(T.data, xform, sizeof T.data);
    Mesh m = ConstructMesh(arr, SurfaceMaterial{}, T);
    return copy_out(m.GetVertices(), m.GetIndices(), v, vcap, idx, icap, icnt);
    MIPT_GUARD_END(0u)
}
uint32_t mipt_builtin_mesh_count(int which) { MIPT_GUARD_BEGIN return (uint32_t)builtin(which).objects.size(); MIPT_GUARD_END(0u) }
uint32_t mipt_builtin_mesh(int which, uint32_t mesh, mipt_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt, mipt_material* mat) {
    MIPT_GUARD_BEGIN
    const Scene& s = builtin(which);
    if (mesh >= s.objects.size()) return 0;
    const Mesh& m = s.objects[mesh];
    if (mat) { memset(mat, 0, sizeof(mipt_material)); memcpy(mat, &m.GetMat(), 45); }   // 45 = through `translucent`; padding stays 0
    return copy_out(m.GetVertices(), m.GetIndices(), v, vcap, idx, icap, icnt);
    MIPT_GUARD_END(0u)
}
void mipt_builtin_sky(int which, mipt_skylight* sky) { try { memcpy(sky, &builtin(which).skyLight, sizeof(mipt_skylight)); } catch (...) { memset(sky, 0, sizeof(mipt_skylight)); } }
void mipt_make_camera(const float pos[3], float azimuth, float zenith, float fov, mipt_camera* out) {
    // (no allocation: plain arithmetic)
    Scenes::CameraParams p; p.pos = { pos[0], pos[1], pos[2] }; p.azimuth = azimuth; p.zenith = zenith; p.fov = fov;
    Camera c; Scenes::UpdateCam(p, c);
    memcpy(out, &c, sizeof(mipt_camera));
}
void mipt_make_synthetic_sky(uint32_t w, uint32_t h, uint32_t seed, float* rgb_out) {
    try {
        std::vector<float> img = Scenes::SyntheticSky(w, h, seed);
        memcpy(rgb_out, img.data(), img.size() * sizeof(float));
    } catch (...) { memset(rgb_out, 0, (size_t)w * h * 3 * sizeof(float)); }
}
uint32_t mipt_make_grid_mesh(uint32_t n, float extent, float height, uint32_t seed, mipt_vertex* v, uint32_t* idx) {
    if (!v || !idx) return (n + 1) * (n + 1);
    MIPT_GUARD_BEGIN
    MeshArrays arr = Scenes::DisplacedGrid(n, extent, height, seed);
    const auto& vv = std::get<0>(arr); const auto& ee = std::get<1>(arr);
    memcpy(v, vv.data(), vv.size() * sizeof(Vertex));
    memcpy(idx, ee.data(), ee.size() * 4);
    return (uint32_t)vv.size();
    MIPT_GUARD_END(0u)
}

int mipt_rgbe_decode(const unsigned char* data, size_t size, float* rgb_out, uint32_t* width, uint32_t* height) {
    if (!data || !width || !height) return MIPT_E_INVALID;
    try {
        std::vector<float> rgb; int w = 0, h = 0;
        if (!Rgbe::Decode(data, size, rgb, w, h)) return MIPT_E_INVALID;
        *width = (uint32_t)w; *height = (uint32_t)h;
        if (rgb_out) memcpy(rgb_out, rgb.data(), rgb.size() * sizeof(float));
        return MIPT_OK;
    } catch (const std::bad_alloc&) { return MIPT_E_NOMEM; } catch (...) { return MIPT_E_INVALID; }
}
size_t mipt_rgbe_encode(const float* rgb, uint32_t width, uint32_t height, unsigned char* out, size_t cap) {
    if (!rgb || !width || !height) return 0;
    MIPT_GUARD_BEGIN
    std::vector<unsigned char> o = Rgbe::Encode(rgb, (int)width, (int)height);
    if (out && cap >= o.size()) memcpy(out, o.data(), o.size());
    return o.size();
    MIPT_GUARD_END((size_t)0)
}

}  // extern "C"
