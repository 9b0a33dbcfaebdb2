// host_sanitize.cpp -- the host-only parts of libmipt under AddressSanitizer + UndefinedBehaviorSanitizer (CPU, no GPU):
// the BVH builders / 4-wide collapse / validation / refit on awkward inputs, and the three file-format parsers
// (Radiance RGBE, Wavefront OBJ, Stanford PLY) on valid, truncated and randomly damaged input.  A parser may refuse
// (return false / throw std::runtime_error); what it may not do is read or write out of bounds, overflow a signed integer
// or leak.  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -static-libasan ...
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../cudapathtracer_amd/csrc/bvh_builder.h"
#include "../cudapathtracer_amd/cpp/ObjLoader.hpp"
#include "../cudapathtracer_amd/cpp/PlyLoader.hpp"
#include "../cudapathtracer_amd/cpp/Rgbe.hpp"
#include "../cudapathtracer_amd/cpp/Scenes.hpp"

using namespace mipt;
namespace Rgbe = CUDATracer::Rgbe;

static int g_checks = 0;
#define REQUIRE(c) do { g_checks++; if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::vector<TriPos> soup(std::mt19937& rng, size_t n, float spread, float size) {
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<TriPos> t(n);
    for (auto& tp : t) {
        const float c[3] = { U(rng) * spread, U(rng) * spread, U(rng) * spread };
        for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) tp.v[k][a] = c[a] + U(rng) * size;
    }
    return t;
}

static int build_and_check(const std::vector<TriPos>& tris, uint32_t builder, uint32_t width, bool mayFail = false) {
    BvhParams p; p.builder = builder; p.nodeWidth = width; p.triCost = width == 2 ? 1.5f : 3.0f;
    BvhOut out;
    const std::string err = build_bvh(tris, p, out);
    if (!err.empty()) {
        if (!mayFail) std::fprintf(stderr, "build_bvh(%zu triangles, builder %u, width %u): %s\n", tris.size(), builder, width, err.c_str());
        REQUIRE(mayFail); return 0;
    }
    REQUIRE(out.nodes.size() % 16 == 0 && out.nodes.size() >= 16);
    BvhOut chk = out;
    REQUIRE(validate_bvh(chk, (uint32_t)tris.size()).empty());
    REQUIRE(chk.stackNeed <= out.stackNeed + 0u || chk.stackNeed == out.stackNeed);
    // refit to moved triangles keeps the tree valid
    std::vector<TriPos> moved = tris;
    for (auto& tp : moved) for (int k = 0; k < 3; k++) { tp.v[k][0] += 0.25f; tp.v[k][1] *= 1.5f; }
    REQUIRE(refit_bvh(out, moved).empty());
    BvhOut chk2 = out;
    REQUIRE(validate_bvh(chk2, (uint32_t)tris.size()).empty());
    // a refit with another triangle count is refused and leaves the tree alone
    if (!moved.empty()) { moved.pop_back(); BvhOut keep = out; REQUIRE(!refit_bvh(out, moved).empty()); REQUIRE(out.nodes.size() == keep.nodes.size() && !std::memcmp(out.nodes.data(), keep.nodes.data(), keep.nodes.size() * 4)); }
    return 0;
}

static int test_builders() {
    std::mt19937 rng(7);
    for (uint32_t builder = 0; builder < 2; builder++) for (uint32_t width : { 0u, 2u, 4u }) {
        for (size_t n : { (size_t)0, (size_t)1, (size_t)2, (size_t)3, (size_t)5, (size_t)64, (size_t)65, (size_t)700 })
            if (build_and_check(soup(rng, n, 4.f, 0.3f), builder, width)) return 1;
        // every triangle the same; degenerate (zero-area) triangles; a needle; far-apart clusters; large triangles that overlap
        std::vector<TriPos> same(300, soup(rng, 1, 1.f, 1.f)[0]);
        if (build_and_check(same, builder, width, /*mayFail: an unsplittable pile makes a deep chain; explicit 4-wide refuses it, auto falls back to BVH2*/ width == 4)) return 1;
        std::vector<TriPos> degen(200); std::memset(degen.data(), 0, degen.size() * sizeof(TriPos));
        if (build_and_check(degen, builder, width)) return 1;
        std::vector<TriPos> needle = soup(rng, 500, 0.001f, 0.0005f);
        for (size_t i = 0; i < needle.size(); i++) for (int k = 0; k < 3; k++) needle[i].v[k][0] += (float)i * 100.f;
        if (build_and_check(needle, builder, width)) return 1;
        std::vector<TriPos> clusters = soup(rng, 600, 1e-3f, 1e-4f);
        for (size_t i = 0; i < clusters.size(); i++) for (int k = 0; k < 3; k++) clusters[i].v[k][i % 3] += (i % 2) ? 1.0e6f : -1.0e6f;
        if (build_and_check(clusters, builder, width, /*mayFail: extent beyond the quantiser's range*/ width == 4)) return 1;
        if (build_and_check(soup(rng, 400, 1.f, 5.f), builder, width)) return 1;
    }
    // damaged flat BVHs are refused, never walked out of bounds
    BvhParams p; BvhOut out;
    REQUIRE(build_bvh(soup(rng, 500, 2.f, 0.2f), p, out).empty());
    std::uniform_int_distribution<size_t> pick(0, out.nodes.size() - 1);
    for (int it = 0; it < 300; it++) {
        BvhOut bad = out;
        const size_t at = pick(rng);
        uint32_t w = (uint32_t)rng(); std::memcpy(&bad.nodes[at], &w, 4);
        (void)validate_bvh(bad, 500);
        g_checks++;
    }
    for (int it = 0; it < 50; it++) { BvhOut bad = out; bad.refTri[rng() % bad.refTri.size()] = (uint32_t)rng(); (void)validate_bvh(bad, 500); g_checks++; }
    { BvhOut bad = out; bad.nodes.resize(bad.nodes.size() - 16); (void)validate_bvh(bad, 500); bad.nodes.clear(); REQUIRE(!validate_bvh(bad, 500).empty()); }
    return 0;
}

template <typename F> static void fuzz(std::mt19937& rng, const std::string& good, int rounds, F&& parse) {
    for (int it = 0; it < rounds; it++) {
        std::string s = good;
        const int mode = it % 4;
        if (mode == 0) s.resize(rng() % (s.size() + 1));                                                        // truncated
        else if (mode == 1) { for (int k = 0; k < 1 + (int)(rng() % 8); k++) s[rng() % s.size()] = (char)rng(); }   // damaged bytes
        else if (mode == 2) { const size_t a = rng() % s.size(); s.insert(a, s.substr(rng() % s.size(), rng() % 64)); }  // spliced
        else { const size_t a = rng() % s.size(); s.erase(a, rng() % 32); }                                      // bytes lost
        try { parse(s); } catch (const std::runtime_error&) {} catch (const std::bad_alloc&) {} catch (const std::length_error&) {}
        g_checks++;
    }
}

static int test_rgbe() {
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> U(0.f, 40.f);
    for (auto wh : { std::pair<int, int>{ 1, 1 }, { 7, 3 }, { 8, 8 }, { 300, 5 }, { 40000, 1 } }) {      // widths below 8 and above 32767 go flat, others RLE
        std::vector<float> img((size_t)wh.first * wh.second * 3);
        for (auto& f : img) f = U(rng) < 2.f ? 0.f : U(rng);
        std::vector<unsigned char> enc = Rgbe::Encode(img.data(), wh.first, wh.second);
        std::vector<float> dec; int w = 0, h = 0;
        REQUIRE(Rgbe::Decode(enc.data(), enc.size(), dec, w, h) && w == wh.first && h == wh.second && dec.size() == img.size());
        const std::string good((const char*)enc.data(), enc.size());
        fuzz(rng, good, 400, [&](const std::string& s) { std::vector<float> d; int a, b; (void)Rgbe::Decode((const unsigned char*)s.data(), s.size(), d, a, b); });
    }
    const char* hostile[] = { "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2147483647 +X 2147483647\n", "#?RADIANCE\n\n-Y -5 +X 9\nabc", "#?RADIANCE\n\n-Y 1 +X 99999999999999999999\n", "", "\n\n\n" };
    for (const char* t : hostile) { std::vector<float> d; int a, b; REQUIRE(!Rgbe::Decode((const unsigned char*)t, std::strlen(t), d, a, b)); }
    return 0;
}

static int test_obj_ply() {
    std::mt19937 rng(13);
    const std::string obj =
        "# cube corner\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvn 0 1 0\nvt 0 0\nvt 1 1\n"
        "f 1/1/1 2/2/1 3/1/1 4/2/1\nf 1//2 2//2 5//2\nf -1 -2 -3\nf 1/1 2/2 3/1\ng part\nusemtl m\ns off\nf 1 2 3 4 5\n";
    { auto m = CUDATracer::ParseOBJ(obj.data(), obj.size()); REQUIRE(std::get<1>(m).size() % 3 == 0 && !std::get<0>(m).empty()); }
    fuzz(rng, obj, 1500, [&](const std::string& s) { auto m = CUDATracer::ParseOBJ(s.data(), s.size()); for (uint32_t i : std::get<1>(m)) if (i >= std::get<0>(m).size()) throw std::logic_error("index out of range"); });
    const char* hostileObj[] = { "f 1 2 3\n", "v 1 2\nf 1 1 1\n", "v 0 0 0\nf 4294967296 1 1\n", "v 0 0 0\nf 0 0 0\n", "v 0 0 0\nf -99999999999 1 1\n", "v nan inf -inf\nv 1 1 1\nv 2 2 2\nf 1 2 3\n" };
    for (const char* t : hostileObj) { try { auto m = CUDATracer::ParseOBJ(t, std::strlen(t)); for (uint32_t i : std::get<1>(m)) REQUIRE(i < std::get<0>(m).size()); } catch (const std::runtime_error&) {} g_checks++; }

    const std::string plyAscii =
        "ply\nformat ascii 1.0\ncomment x\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
        "element face 2\nproperty list uchar int vertex_indices\nend_header\n0 0 0 0 0 1\n1 0 0 0 0 1\n1 1 0 0 0 1\n0 1 0 0 0 1\n3 0 1 2\n4 0 1 2 3\n";
    { auto m = CUDATracer::ParsePLY(plyAscii.data(), plyAscii.size()); REQUIRE(std::get<0>(m).size() == 4 && std::get<1>(m).size() == 9); }
    fuzz(rng, plyAscii, 1500, [&](const std::string& s) { auto m = CUDATracer::ParsePLY(s.data(), s.size()); for (uint32_t i : std::get<1>(m)) if (i >= std::get<0>(m).size()) throw std::logic_error("index out of range"); });
    std::string plyBin = "ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                         "element face 1\nproperty list uchar uint vertex_indices\nend_header\n";
    const float vb[9] = { 0, 0, 0, 1, 0, 0, 0, 1, 0 };
    for (int v = 0; v < 3; v++) { plyBin.append((const char*)&vb[v * 3], 12); plyBin.push_back((char)200); }
    plyBin.push_back((char)3); const uint32_t idx[3] = { 0, 1, 2 }; plyBin.append((const char*)idx, 12);
    { auto m = CUDATracer::ParsePLY(plyBin.data(), plyBin.size()); REQUIRE(std::get<0>(m).size() == 3 && std::get<1>(m).size() == 3); }
    fuzz(rng, plyBin, 1500, [&](const std::string& s) { auto m = CUDATracer::ParsePLY(s.data(), s.size()); for (uint32_t i : std::get<1>(m)) if (i >= std::get<0>(m).size()) throw std::logic_error("index out of range"); });
    const char* hostilePly[] = { "ply\nformat ascii 1.0\nelement vertex 4000000000\nproperty float x\nproperty float y\nproperty float z\nend_header\n",
                                 "ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nelement face 1000000000\nproperty list uchar int vertex_indices\nend_header\n\0\0\0\0\0\0\0\0\0\0\0\0",
                                 "ply\nformat ascii 1.0\nelement vertex 1\nproperty list uchar float x\nend_header\n255 1 2 3\n", "ply\n", "" };
    for (const char* t : hostilePly) { try { (void)CUDATracer::ParsePLY(t, std::strlen(t)); } catch (const std::runtime_error&) {} catch (const std::bad_alloc&) {} catch (const std::length_error&) {} g_checks++; }
    return 0;
}

static int test_scenes() {
    using namespace CUDATracer;
    Scene a = Scenes::CornellBox(), b = Scenes::DefaultScene();
    size_t tris = 0; for (const Mesh& m : a.objects) tris += m.GetIndices().size() / 3;
    REQUIRE(tris == 36 && b.objects.size() == 5);
    for (const Scene* s : { &a, &b }) {
        std::vector<TriPos> t;
        for (const Mesh& m : s->objects) { const auto& v = m.GetVertices(); const auto& e = m.GetIndices();
            for (size_t i = 0; i + 2 < e.size(); i += 3) { TriPos tp; for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) tp.v[k][c] = v[e[i + k]].position[c]; t.push_back(tp); } }
        for (uint32_t width : { 0u, 2u, 4u }) if (build_and_check(t, 0, width)) return 1;
    }
    return 0;
}

#include <chrono>
int main() {
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { auto t1 = std::chrono::steady_clock::now(); std::fprintf(stderr, "  %s: %.1f s\n", what, std::chrono::duration<double>(t1 - t0).count()); t0 = t1; };
    if (test_builders()) return 1; lap("builders");
    if (test_rgbe()) return 1; lap("rgbe");
    if (test_obj_ply()) return 1; lap("obj/ply");
    if (test_scenes()) return 1; lap("scenes");
    std::printf("host_sanitize: %d checks, no sanitizer report\n", g_checks);
    return 0;
}
