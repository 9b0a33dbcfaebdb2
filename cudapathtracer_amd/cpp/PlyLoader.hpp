// PlyLoader.hpp -- Stanford PLY ingestion for the host scene types (beside ObjLoader.hpp).
// The reference has no mesh loader (its scenes are procedural, Types.hpp:435-557); LoadPLY returns the same
// MeshArrays tuple as MakeCube/MakeSphere/MakeQuad, so ConstructMesh(LoadPLY(path), material, transform) drops in.
//
// Supported: format ascii 1.0 and binary_little_endian 1.0; element vertex with scalar properties x y z (required),
// nx ny nz and s t / u v (optional), any other scalar or list property skipped; element face with one list property
// (vertex_indices / vertex_index) of any integer types, polygons fan-triangulated (counter-clockwise = front); other
// elements are skipped.  Vertices without normals get the area-weighted average of the normals of their polygons.
// binary_big_endian is rejected.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Types.hpp"

namespace CUDATracer {
namespace detail {

struct PlyProp { std::string name; int type = 0; bool list = false; int countType = 0; };   // type: index into kPlyTypes
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };

inline int ply_type(const std::string& t) {
    static const char* names[][2] = { {"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                      {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"} };
    for (int i = 0; i < 8; i++) if (t == names[i][0] || t == names[i][1]) return i;
    throw std::runtime_error("PLY: unknown property type " + t);
}
inline size_t ply_size(int type) { static const size_t s[8] = { 1, 1, 2, 2, 4, 4, 4, 8 }; return s[type]; }

// cursor over the body: reads one scalar of the given type as double, ascii or little-endian binary
struct PlyCursor {
    const char* p; const char* end; bool ascii;
    double scalar(int type) {
        if (ascii) {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) p++;
            if (p >= end) throw std::runtime_error("PLY: unexpected end of data");
            char* e; double v = std::strtod(p, &e);
            if (e == p) throw std::runtime_error("PLY: malformed number");
            p = e; return v;
        }
        const size_t n = ply_size(type);
        if ((size_t)(end - p) < n) throw std::runtime_error("PLY: unexpected end of data");
        double v = 0;
        switch (type) {
            case 0: { int8_t x; std::memcpy(&x, p, 1); v = x; break; }
            case 1: { uint8_t x; std::memcpy(&x, p, 1); v = x; break; }
            case 2: { int16_t x; std::memcpy(&x, p, 2); v = x; break; }
            case 3: { uint16_t x; std::memcpy(&x, p, 2); v = x; break; }
            case 4: { int32_t x; std::memcpy(&x, p, 4); v = x; break; }
            case 5: { uint32_t x; std::memcpy(&x, p, 4); v = x; break; }
            case 6: { float x; std::memcpy(&x, p, 4); v = x; break; }
            default: { double x; std::memcpy(&x, p, 8); v = x; break; }
        }
        p += n; return v;
    }
};

}  // namespace detail

inline MeshArrays ParsePLY(const char* data, size_t len) {
    using namespace detail;
    // ---- header (always text, ends with "end_header\n")
    const char* p = data; const char* end = data + len;
    auto next_line = [&]() -> std::string {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) throw std::runtime_error("PLY: header not terminated");
        std::string l(p, eol); p = eol + 1;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        return l;
    };
    if (next_line() != "ply") throw std::runtime_error("PLY: missing magic");
    bool ascii = false, haveFormat = false;
    std::vector<PlyElem> elems;
    while (true) {
        std::istringstream ls(next_line());
        std::string kw; ls >> kw;
        if (kw == "end_header") break;
        if (kw == "comment" || kw == "obj_info" || kw.empty()) continue;
        if (kw == "format") {
            std::string f, ver; ls >> f >> ver;
            if (f == "ascii") ascii = true;
            else if (f == "binary_little_endian") ascii = false;
            else throw std::runtime_error("PLY: unsupported format " + f);
            haveFormat = true;
        } else if (kw == "element") {
            PlyElem e; ls >> e.name >> e.count;
            if (ls.fail()) throw std::runtime_error("PLY: malformed element line");
            elems.push_back(e);
        } else if (kw == "property") {
            if (elems.empty()) throw std::runtime_error("PLY: property before element");
            PlyProp pr; std::string t; ls >> t;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> pr.name; pr.list = true; pr.countType = ply_type(ct); pr.type = ply_type(it); }
            else { pr.type = ply_type(t); ls >> pr.name; }
            if (pr.name.empty()) throw std::runtime_error("PLY: malformed property line");
            elems.back().props.push_back(pr);
        } else throw std::runtime_error("PLY: unknown header keyword " + kw);
    }
    if (!haveFormat) throw std::runtime_error("PLY: no format line");

    // ---- body
    PlyCursor cur{ p, end, ascii };
    std::vector<Vertex> verts; std::vector<std::uint32_t> idx;
    std::vector<std::vector<std::uint32_t>> polys;
    bool haveNormals = false;
    // list length as a count: finite, non-negative, and no longer than the data that is left could hold
    auto list_count = [&](const PlyProp& pr) -> size_t {
        const double c = cur.scalar(pr.countType);
        if (!(c >= 0.0) || c > (double)(cur.end - cur.p)) throw std::runtime_error("PLY: list length out of range");
        return (size_t)c;
    };
    for (const PlyElem& e : elems) {
        // an element count that the remaining bytes cannot hold is refused before anything is allocated or looped over:
        // a row takes at least the sizes of its scalars (binary) or a digit and a separator per value (ascii)
        size_t minRow = 0;
        for (const PlyProp& pr : e.props) minRow += ascii ? 2 : ply_size(pr.list ? pr.countType : pr.type);
        if (minRow == 0) { if (e.name == "vertex") throw std::runtime_error("PLY: vertex element without x y z"); continue; }
        if (e.count > (size_t)(cur.end - cur.p) / minRow + 1) throw std::runtime_error("PLY: unexpected end of data (element " + e.name + " claims more rows than the bytes left can hold)");
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t k = 0; k < e.props.size(); k++) {
                const std::string& n = e.props[k].name;
                if (e.props[k].list) continue;
                if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
                else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
                else if (n == "s" || n == "u") iu = (int)k; else if (n == "t" || n == "v") iv = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("PLY: vertex element without x y z");
            haveNormals = inx >= 0 && iny >= 0 && inz >= 0;
            verts.resize(e.count);
            std::vector<double> val(e.props.size());
            for (size_t i = 0; i < e.count; i++) {
                for (size_t k = 0; k < e.props.size(); k++) {
                    const PlyProp& pr = e.props[k];
                    if (pr.list) { const size_t c = list_count(pr); for (size_t j = 0; j < c; j++) cur.scalar(pr.type); val[k] = 0; }
                    else val[k] = cur.scalar(pr.type);
                }
                Vertex& vx = verts[i];
                vx.position = { (float)val[ix], (float)val[iy], (float)val[iz] };
                vx.normal = haveNormals ? Math::vec3f{ (float)val[inx], (float)val[iny], (float)val[inz] } : Math::vec3f{ 0, 0, 0 };
                vx.uv = (iu >= 0 && iv >= 0) ? Math::vec2f{ (float)val[iu], (float)val[iv] } : Math::vec2f{ 0, 0 };
            }
        } else {
            const bool face = e.name == "face";
            for (size_t i = 0; i < e.count; i++) {
                for (const PlyProp& pr : e.props) {
                    if (!pr.list) { cur.scalar(pr.type); continue; }
                    const size_t c = list_count(pr);
                    const bool isIdx = face && (pr.name == "vertex_indices" || pr.name == "vertex_index");
                    std::vector<std::uint32_t> poly;
                    for (size_t j = 0; j < c; j++) {
                        const double v = cur.scalar(pr.type);
                        if (isIdx) {
                            if (v < 0 || v >= (double)verts.size()) throw std::runtime_error("PLY: vertex index out of range in face " + std::to_string(i));
                            poly.push_back((std::uint32_t)v);
                        }
                    }
                    if (isIdx) {
                        if (poly.size() < 3) throw std::runtime_error("PLY: face with fewer than 3 corners");
                        polys.push_back(std::move(poly));
                    }
                }
            }
        }
    }
    for (const auto& poly : polys)
        for (size_t k = 1; k + 1 < poly.size(); k++) { idx.push_back(poly[0]); idx.push_back(poly[k]); idx.push_back(poly[k + 1]); }
    if (!haveNormals) {
        for (const auto& poly : polys) {
            Math::vec3f fn{ 0, 0, 0 };
            for (size_t k = 1; k + 1 < poly.size(); k++)
                fn = fn + Math::cross(verts[poly[k]].position - verts[poly[0]].position, verts[poly[k + 1]].position - verts[poly[0]].position);
            for (std::uint32_t v : poly) verts[v].normal = verts[v].normal + fn;
        }
        for (Vertex& v : verts) {
            float l = std::sqrt(Math::dot(v.normal, v.normal));
            v.normal = l > 0.f ? v.normal * (1.f / l) : Math::vec3f{ 0, 1, 0 };
        }
    }
    return std::make_tuple(verts, idx);
}

inline MeshArrays LoadPLY(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string data; char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    std::fclose(f);
    return ParsePLY(data.data(), data.size());
}

}  // namespace CUDATracer
