// Types.hpp -- host scene types with the reference's names and field order
// (reference: CUDATracer/Types.hpp:20-39,214-231,272-310,427-574).  The BVH is NOT built here: the reference builds a
// median BVH inside Mesh's constructor; this tree hands the triangles to libmipt, which builds one flat BVH per scene.
#pragma once
#include <cstdint>
#include <tuple>
#include <vector>

#include "Math.hpp"

namespace CUDATracer {

struct Camera { Math::vec3f pos, front, up, right; float fov = 0; };                       // 52 B
struct Vertex { Math::vec3f position, normal; Math::vec2f uv; };                           // 32 B
struct SurfaceMaterial {                                                                    // 48 B
    Math::vec3f color; float roughness = 0, metallic = 0, opacity = 0;
    Math::vec3f emissiveColor; float emissiveStrength = 0; float n = 0; bool translucent = false;
};
struct SkyLight { Math::vec3f color, dir; float intensity = 0; };                           // 28 B
struct AABB { Math::vec3f min{ INFINITY }, max{ -INFINITY }; };

using MeshArrays = std::tuple<std::vector<Vertex>, std::vector<std::uint32_t>>;

class Mesh {
    std::vector<Vertex> _vert; std::vector<std::uint32_t> _ind; AABB _aabb; SurfaceMaterial _mat;
public:
    Math::mat4x4f transform{ 1.f };   // kept for API parity; vertices are already transformed (ConstructMesh bakes it)
    Mesh(const std::vector<Vertex>& vert, const std::vector<std::uint32_t>& ind, const SurfaceMaterial& mat)
        : _vert(vert), _ind(ind), _mat(mat) {
        for (const Vertex& v : _vert) for (int a = 0; a < 3; a++) {
            if (v.position[a] < _aabb.min[a]) _aabb.min[a] = v.position[a];
            if (v.position[a] > _aabb.max[a]) _aabb.max[a] = v.position[a];
        }
    }
    const AABB GetAABB() const { return _aabb; }
    const SurfaceMaterial& GetMat() const { return _mat; }
    const std::vector<Vertex>& GetVertices() const { return _vert; }
    const std::vector<std::uint32_t>& GetIndices() const { return _ind; }
};

struct Scene {
    SkyLight skyLight;
    std::vector<float> envMap;      // float RGB rows, top row = +Y
    Math::vec2ui envMapDim;
    std::vector<Mesh> objects;
};

namespace detail {
inline void quad_indices(std::vector<std::uint32_t>& e, std::uint32_t a, std::uint32_t b, std::uint32_t c, std::uint32_t d) {
    const std::uint32_t w[6] = { a, b, c, a, c, d }; e.insert(e.end(), w, w + 6);
}
}  // namespace detail

// unit cube [0,1]^3, 24 vertices / 12 triangles, face order -y +y +x -x +z -z (Types.hpp:435-470)
inline MeshArrays MakeCube(const Math::vec3f& offset = {}) {
    std::vector<Vertex> v; std::vector<std::uint32_t> e;
    struct Face { int axis; float val; Math::vec3f n; };
    const Face faces[6] = { {1, 0, {0, -1, 0}}, {1, 1, {0, 1, 0}}, {0, 1, {1, 0, 0}}, {0, 0, {-1, 0, 0}}, {2, 1, {0, 0, 1}}, {2, 0, {0, 0, -1}} };
    const float ab[4][2] = { {0, 0}, {1, 0}, {1, 1}, {0, 1} };
    for (const Face& f : faces) for (int i = 0; i < 4; i++) {
        float a = ab[i][0], b = ab[i][1];
        Math::vec3f p = f.axis == 0 ? Math::vec3f(f.val, a, b) : f.axis == 1 ? Math::vec3f(a, f.val, b) : Math::vec3f(a, b, f.val);
        v.push_back({ p + offset, f.n, Math::vec2f(a, b) });
    }
    detail::quad_indices(e, 0, 1, 2, 3);     detail::quad_indices(e, 7, 6, 5, 4);
    detail::quad_indices(e, 8, 9, 10, 11);   detail::quad_indices(e, 15, 14, 13, 12);
    detail::quad_indices(e, 16, 17, 18, 19); detail::quad_indices(e, 23, 22, 21, 20);
    return std::make_tuple(v, e);
}

// unit sphere, 40 x 80 segments, 3122 vertices / 6240 triangles; ring uv is 0 as in the reference (Types.hpp:473-538)
inline MeshArrays MakeSphere() {
    const unsigned seg_y = 40, seg_x = 80;
    std::vector<Vertex> v{ { {0, 1, 0}, {0, 1, 0}, {0.5f, 1} }, { {0, -1, 0}, {0, -1, 0}, {0.5f, 0} } };
    std::vector<std::uint32_t> e;
    const float delta_x = (float)(M_PI * 2 / seg_x), delta_y = (float)(M_PI / seg_y);
    for (int y = 1; y < (int)seg_y; y++) {
        float zenith = y * delta_y, r = std::sin(zenith), height = -std::cos(zenith);
        for (int x = 0; x < (int)seg_x; x++) {
            float az = x * delta_x, cx = std::cos(az) * r, cz = std::sin(az) * r;
            v.push_back({ {cx, height, cz}, {cx, height, cz}, {0.f, 0.f} });
        }
    }
    for (int y = 0; y < (int)seg_y - 2; y++) for (int x = 0; x < (int)seg_x; x++) {
        unsigned nx = (x + 1) % seg_x, ny = y + 1;
        detail::quad_indices(e, 2 + y * seg_x + nx, 2 + y * seg_x + x, 2 + ny * seg_x + x, 2 + ny * seg_x + nx);
    }
    const std::uint32_t top = (std::uint32_t)v.size() - seg_x, bottom = 2;
    for (unsigned x = 0; x < seg_x; x++) {
        unsigned nx = (x + 1) % seg_x;
        const std::uint32_t w[6] = { 0, top + nx, top + x, 1, bottom + x, bottom + nx }; e.insert(e.end(), w, w + 6);
    }
    return std::make_tuple(v, e);
}

// quad in the y=0 plane, normal +y (Types.hpp:540-557)
inline MeshArrays MakeQuad(Math::vec2f size = { 1, 1 }, Math::vec3f offset = {}) {
    const Math::vec3f n{ 0, 1, 0 };
    std::vector<Vertex> v{ { Math::vec3f{0, 0, 0} + offset, n, {0, 0} }, { Math::vec3f{0, 0, size.x} + offset, n, {0, 1} },
                           { Math::vec3f{size.y, 0, size.x} + offset, n, {1, 1} }, { Math::vec3f{size.y, 0, 0} + offset, n, {1, 0} } };
    std::vector<std::uint32_t> e; detail::quad_indices(e, 0, 1, 2, 3);
    return std::make_tuple(v, e);
}

// bakes the transform into positions (w=1) and normals (same matrix, w=0, not renormalised) -- Types.hpp:560-574
inline Mesh ConstructMesh(const MeshArrays& pair, const SurfaceMaterial& mat, const Math::mat4x4f& transform) {
    auto vert = std::get<0>(pair);
    for (auto& v : vert) { v.position = Math::MatMul(transform, v.position, 1.f); v.normal = Math::MatMul(transform, v.normal, 0.f); }
    return Mesh(vert, std::get<1>(pair), mat);
}

}  // namespace CUDATracer
