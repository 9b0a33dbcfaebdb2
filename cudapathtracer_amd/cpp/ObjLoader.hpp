// ObjLoader.hpp -- Wavefront OBJ ingestion for the host scene types (the step before the path).
// The reference has no mesh loader: its scenes are procedural (Types.hpp:435-557) and enter through
// Mesh(vertices, indices, material) / ConstructMesh (Types.hpp:283-293,560-574).  LoadOBJ returns the same MeshArrays
// tuple as MakeCube/MakeSphere/MakeQuad, so a file drops into ConstructMesh(LoadOBJ(path), material, transform).
//
// Supported: v / vn / vt / f records with `a`, `a/b`, `a//c`, `a/b/c` corners, negative (relative) indices, polygons
// (fan-triangulated, counter-clockwise = front, as Types.hpp:271), `\` line continuation is not supported.
// Corners without a normal get the area-weighted average of the normals of the polygons around their position.
// Everything else (groups, materials, lines, curves) is ignored; one OBJ file = one mesh with one material.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "Types.hpp"

namespace CUDATracer {

inline MeshArrays ParseOBJ(const char* text, size_t len) {
    std::vector<Math::vec3f> pos, nrm; std::vector<Math::vec2f> tex;
    struct Corner { int v, t, n; };
    std::vector<Corner> corners;            // 3 per triangle
    std::vector<Math::vec3f> smooth;        // per position: area-weighted sum of the normals of the polygons around it
    auto fail = [](const std::string& m, size_t line) -> void { throw std::runtime_error("OBJ line " + std::to_string(line) + ": " + m); };
    size_t lineNo = 0;
    const char* p = text; const char* end = text + len;
    while (p < end) {
        const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        std::string line(p, eol); p = eol + 1; lineNo++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* e; float x = strtof(s + 2, &e), y = strtof(e, &e), z = strtof(e, &e);
            pos.push_back({ x, y, z });
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            char* e; float x = strtof(s + 3, &e), y = strtof(e, &e), z = strtof(e, &e);
            nrm.push_back({ x, y, z });
        } else if (s[0] == 'v' && s[1] == 't' && (s[2] == ' ' || s[2] == '\t')) {
            char* e; float x = strtof(s + 3, &e), y = strtof(e, &e);
            tex.push_back({ x, y });
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<Corner> poly;
            const char* q = s + 2;
            while (true) {
                while (*q == ' ' || *q == '\t') q++;
                if (!*q) break;
                Corner c{ 0, 0, 0 };
                char* e;
                long v = strtol(q, &e, 10);
                if (e == q) fail("bad face corner", lineNo);
                c.v = (int)v; q = e;
                if (*q == '/') {
                    q++;
                    if (*q != '/') { long t = strtol(q, &e, 10); if (e == q) fail("bad texture index", lineNo); c.t = (int)t; q = e; }
                    if (*q == '/') { q++; long n = strtol(q, &e, 10); if (e == q) fail("bad normal index", lineNo); c.n = (int)n; q = e; }
                }
                auto resolve = [&](int idx, size_t count, const char* what) -> int {     // 1-based / negative -> 0-based, -1 = absent
                    if (idx == 0) return -1;
                    long r = idx > 0 ? idx - 1 : (long)count + idx;
                    if (r < 0 || (size_t)r >= count) fail(std::string(what) + " index out of range", lineNo);
                    return (int)r;
                };
                if (c.v == 0) fail("vertex index 0", lineNo);
                c.v = resolve(c.v, pos.size(), "vertex"); c.t = resolve(c.t, tex.size(), "texture"); c.n = resolve(c.n, nrm.size(), "normal");
                poly.push_back(c);
            }
            if (poly.size() < 3) fail("face with fewer than 3 corners", lineNo);
            Math::vec3f fn{ 0, 0, 0 };                            // polygon normal, length = 2 x area
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
                fn = fn + Math::cross(pos[poly[k].v] - pos[poly[0].v], pos[poly[k + 1].v] - pos[poly[0].v]);
            }
            if (smooth.size() < pos.size()) smooth.resize(pos.size(), Math::vec3f{ 0, 0, 0 });
            for (const Corner& c : poly) smooth[c.v] = smooth[c.v] + fn;     // each polygon counts once per corner
        }
    }
    // smooth normals for corners that carry none
    smooth.resize(pos.size(), Math::vec3f{ 0, 0, 0 });
    for (auto& n : smooth) {
        float l = std::sqrt(Math::dot(n, n));
        n = l > 0.f ? n * (1.f / l) : Math::vec3f{ 0, 1, 0 };
    }
    // one Vertex per distinct (position, texture, normal) triple
    std::vector<Vertex> verts; std::vector<std::uint32_t> idx;
    struct Key { int v, t, n; bool operator==(const Key& o) const { return v == o.v && t == o.t && n == o.n; } };
    struct KeyHash { size_t operator()(const Key& k) const { return ((size_t)(unsigned)k.v * 73856093u) ^ ((size_t)(unsigned)k.t * 19349663u) ^ ((size_t)(unsigned)k.n * 83492791u); } };
    std::unordered_map<Key, std::uint32_t, KeyHash> seen;
    idx.reserve(corners.size());
    for (const Corner& c : corners) {
        Key k{ c.v, c.t, c.n };
        auto it = seen.find(k);
        if (it == seen.end()) {
            Vertex vx;
            vx.position = pos[c.v];
            vx.normal = c.n >= 0 ? nrm[c.n] : smooth[c.v];
            vx.uv = c.t >= 0 ? tex[c.t] : Math::vec2f{ 0, 0 };
            it = seen.emplace(k, (std::uint32_t)verts.size()).first;
            verts.push_back(vx);
        }
        idx.push_back(it->second);
    }
    return std::make_tuple(verts, idx);
}

inline MeshArrays LoadOBJ(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string data; char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    std::fclose(f);
    return ParseOBJ(data.data(), data.size());
}

}  // namespace CUDATracer
