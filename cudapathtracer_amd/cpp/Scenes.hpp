// Scenes.hpp -- the two benchmark scenes of the reference restated with the CUDATracer host API, plus the
// synthetic inputs the BASELINE configs need (HDRI stand-in for the missing sky.hdr, displaced-grid stress mesh).
// Reference: CornellBox/tracer.cpp:16-158, DefaultScene/tracer.cpp:28-123, Common/RendererWindow.hpp:15-37,76-85.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "Types.hpp"

namespace CUDATracer {
namespace Scenes {

struct CameraParams { float fov = 0, azimuth = 0, zenith = 0; Math::vec3f pos; };

// RendererWindow::UpdateCam (Common/RendererWindow.hpp:15-37)
inline void UpdateCam(CameraParams& p, Camera& cam) {
    cam.pos = p.pos; cam.fov = p.fov;
    p.zenith = p.zenith < 0.f ? 0.f : (p.zenith > (float)M_PI ? (float)M_PI : p.zenith);
    float fy = std::cos(p.zenith), d = std::sin(p.zenith), fx_ = std::cos(p.azimuth), fz_ = std::sin(p.azimuth);
    Math::vec3f front(fx_ * d, fy, fz_ * d), right(-fz_, 0, fx_);
    cam.front = front; cam.right = right; cam.up = Math::cross(right, front);
}

inline SkyLight DefaultSky(float intensity) {
    return SkyLight{ {0.98f, 0.97f, 0.99f}, -Math::normalize(Math::vec3f{1, 1, 1}), intensity };
}

// Cornell box: floor, ceiling, back, left(red), right(green), light, mirror box, diffuse box; no front wall.
inline Scene CornellBox() {
    using namespace Math;
    SurfaceMaterial diffuse{}; diffuse.color = 1.0f; diffuse.roughness = 0.5f; diffuse.metallic = 0.0f; diffuse.opacity = 1.0f; diffuse.n = 2.5f;
    const mat4x4f wall = MatMul(MakeTranslate(vec3f{-2, -2, -2}), MakeScale(vec3f{4, 1, 4}));
    auto rotated = [&](const vec3f& axis, float rad) { return MatMul(MakeRotation(Quaternion3f::rotate(axis, rad)), wall); };
    Scene scn; scn.skyLight = DefaultSky(0);
    scn.objects.push_back(ConstructMesh(MakeQuad(), diffuse, wall));
    scn.objects.push_back(ConstructMesh(MakeQuad(), diffuse, rotated({0, 0, 1}, (float)M_PI)));
    scn.objects.push_back(ConstructMesh(MakeQuad(), diffuse, rotated({0, 0, 1}, (float)(M_PI / 2))));
    SurfaceMaterial red = diffuse; red.color = {1, 0, 0};
    scn.objects.push_back(ConstructMesh(MakeQuad(), red, rotated({1, 0, 0}, (float)(M_PI / 2))));
    SurfaceMaterial green = diffuse; green.color = {0, 1, 0};
    scn.objects.push_back(ConstructMesh(MakeQuad(), green, rotated({1, 0, 0}, (float)(-M_PI / 2))));
    SurfaceMaterial light = diffuse; light.emissiveColor = {1.f, .8f, 0.5f}; light.emissiveStrength = 10;
    scn.objects.push_back(ConstructMesh(MakeQuad(), light,
        MatMul(MakeRotation(Quaternion3f::rotate({0, 0, 1}, (float)M_PI)), MakeTranslate(vec3f{-0.5f, -1.99999f, -0.5f}))));
    SurfaceMaterial mirror = diffuse; mirror.roughness = 0; mirror.metallic = 1;
    scn.objects.push_back(ConstructMesh(MakeCube(), mirror,
        MatMul(MakeTranslate(vec3f{0.5f - 0.5f, -2, -0.5f - 0.5f}), MatMul(MakeRotation(Quaternion3f::rotate({0, 1, 0}, 0.4f)), MakeScale(vec3f{1, 2.5f, 1})))));
    SurfaceMaterial box2 = diffuse; box2.roughness = 0.5f; box2.metallic = 0;
    scn.objects.push_back(ConstructMesh(MakeCube(), box2,
        MatMul(MakeTranslate(vec3f{-0.5f - 0.5f, -2, 0.5f - 0.5f}), MakeRotation(Quaternion3f::rotate({0, 1, 0}, -0.4f)))));
    return scn;
}
inline CameraParams CornellCamera() { CameraParams p; p.pos = {-6, 0, 0}; p.fov = (float)(M_PI / 3); p.zenith = (float)(M_PI / 2); p.azimuth = 0; return p; }

// DefaultScene: gold cube, glass-blue cube, ground quad, green glass sphere, teal metal sphere.
inline Scene DefaultScene() {
    using namespace Math;
    auto material = [](vec3f c, float rough, float metal, float opacity, float n) {
        SurfaceMaterial m{}; m.color = c; m.roughness = rough; m.metallic = metal; m.opacity = opacity; m.n = n; return m; };
    const SurfaceMaterial gold = material({0.9f, 0.7f, 0.1f}, 0.01f, 0.95f, 1.f, 1.5f);
    const SurfaceMaterial blue = material({0.5f, 0.5f, 0.9f}, 0.0f, 0.1f, 0.2f, 2.7f);
    const SurfaceMaterial greenGlass = material({0.3f, 0.95f, 0.4f}, 0.00009f, 0.01f, 0.2f, 1.7f);
    const SurfaceMaterial teal = material({0.1f, 0.95f, 0.84f}, 0.7f, 0.9f, 0.3f, 1.5f);
    const SurfaceMaterial ground = material({0.9f, 0.9f, 0.8f}, 0.5f, 0.01f, 1.f, 2.7f);
    Scene scn; scn.skyLight = DefaultSky(5);
    scn.objects.push_back(ConstructMesh(MakeCube(), gold, MakeTranslate(vec3f(0, 0.1f, -0.5f))));
    scn.objects.push_back(ConstructMesh(MakeCube(), blue,
        MatMul(MakeTranslate(vec3f(1.3f, 0, 1.7f)), MatMul(MakeRotation(Quaternion3f::rotate(vec3f(0, 1, 0), 0.2f)), MakeScale(vec3f(1.2f, 1.2f, 1.2f))))));
    scn.objects.push_back(ConstructMesh(MakeQuad(), ground, MatMul(MakeTranslate(vec3f(-4, 0, -4)), MakeScale(vec3f{8.f, 1.f, 8.f}))));
    scn.objects.push_back(ConstructMesh(MakeSphere(), greenGlass, MatMul(MakeTranslate(vec3f(-1.5f, 0.5f, 1.2f)), MakeScale(vec3f{.5f, .5f, .5f}))));
    scn.objects.push_back(ConstructMesh(MakeSphere(), teal, MatMul(MakeTranslate(vec3f(-1.2f, 1, -1.2f)), MakeScale(vec3f{.8f, .8f, .8f}))));
    return scn;
}
inline CameraParams DefaultCamera() { CameraParams p; p.pos = {-1, 1, 0}; p.fov = (float)(M_PI / 3); p.zenith = (float)(M_PI / 2); p.azimuth = 0; return p; }

// ---- synthetic inputs ------------------------------------------------------------------------------------------
namespace detail {
inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
inline float lattice(int x, int y, uint32_t seed) { return (hash32((uint32_t)x * 0x9E3779B9u ^ hash32((uint32_t)y + seed)) & 0xFFFFFF) / (float)0x1000000; }
inline float value_noise(float x, float y, uint32_t seed) {
    float fx = std::floor(x), fy = std::floor(y); int ix = (int)fx, iy = (int)fy;
    float tx = x - fx, ty = y - fy; tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty);
    float a = lattice(ix, iy, seed), b = lattice(ix + 1, iy, seed), c = lattice(ix, iy + 1, seed), d = lattice(ix + 1, iy + 1, seed);
    float ab = a + (b - a) * tx, cd = c + (d - c) * tx;
    return ab + (cd - ab) * ty;
}
inline float fbm4(float x, float y, uint32_t seed) {
    float s = 0, amp = 0.5f;
    for (int o = 0; o < 4; o++) { s += amp * value_noise(x, y, seed + (uint32_t)o * 101u); x *= 2; y *= 2; amp *= 0.5f; }
    return s;
}
}  // namespace detail

// Stand-in for DefaultScene/sky.hdr (absent from the reference checkout): equirect float RGB, top row = +Y.
// Sky gradient above the horizon, a 3-degree sun of radiance 5e4 towards normalize(1,1,1), value-noise ground below.
inline std::vector<float> SyntheticSky(uint32_t w, uint32_t h, uint32_t seed) {
    std::vector<float> img((size_t)w * h * 3);
    const Math::vec3f sun = Math::normalize(Math::vec3f{1, 1, 1});
    const float cosSun = std::cos(3.0f * (float)M_PI / 180.0f);
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
        float zen = ((float)y + 0.5f) / (float)h * (float)M_PI, az = (((float)x + 0.5f) / (float)w - 0.5f) * 2.0f * (float)M_PI;
        Math::vec3f d(std::sin(zen) * std::sin(az), std::cos(zen), std::sin(zen) * std::cos(az));   // inverse of the lookup in OnMiss
        Math::vec3f c;
        if (d.y >= 0) {
            float t = d.y; c = Math::vec3f(0.9f, 0.95f, 1.0f) * (1 - t) + Math::vec3f(0.25f, 0.45f, 0.9f) * t;
            if (Math::dot(d, sun) > cosSun) c = Math::vec3f(5.0e4f, 4.8e4f, 4.2e4f);
        } else {
            float g = 0.02f + 0.38f * detail::fbm4((float)x * 64.0f / (float)w, (float)y * 64.0f / (float)h, seed);
            c = Math::vec3f(g * 0.8f, g, g * 0.6f);
        }
        size_t a = ((size_t)y * w + x) * 3; img[a] = c.x; img[a + 1] = c.y; img[a + 2] = c.z;
    }
    return img;
}

// n x n quads spanning [-extent, extent]^2, heights = 4-octave value noise * height; shared vertices, smooth normals.
inline MeshArrays DisplacedGrid(uint32_t n, float extent, float height, uint32_t seed) {
    std::vector<Vertex> v((size_t)(n + 1) * (n + 1)); std::vector<std::uint32_t> e; e.reserve((size_t)n * n * 6);
    auto H = [&](int i, int j) { return height * detail::fbm4((float)i * 24.0f / (float)n, (float)j * 24.0f / (float)n, seed); };
    const float step = 2 * extent / (float)n;
    for (uint32_t j = 0; j <= n; j++) for (uint32_t i = 0; i <= n; i++) {
        float hx0 = H((int)i - 1, (int)j), hx1 = H((int)i + 1, (int)j), hz0 = H((int)i, (int)j - 1), hz1 = H((int)i, (int)j + 1);
        Vertex& p = v[(size_t)j * (n + 1) + i];
        p.position = Math::vec3f(-extent + step * (float)i, H((int)i, (int)j), -extent + step * (float)j);
        p.normal = Math::normalize(Math::vec3f(hx0 - hx1, 2 * step, hz0 - hz1));
        p.uv = Math::vec2f((float)i / (float)n, (float)j / (float)n);
    }
    for (uint32_t j = 0; j < n; j++) for (uint32_t i = 0; i < n; i++) {
        std::uint32_t a = j * (n + 1) + i, b = a + 1, c = a + (n + 1), d = c + 1;
        const std::uint32_t w[6] = { a, c, d, a, d, b }; e.insert(e.end(), w, w + 6);   // counter-clockwise seen from +y
    }
    return std::make_tuple(v, e);
}

}  // namespace Scenes
}  // namespace CUDATracer
