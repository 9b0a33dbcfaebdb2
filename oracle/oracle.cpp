/*
 * oracle.cpp -- scalar CPU restatement of the reference path tracer.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity is pinned by images only: the reference holds no golden vectors for
 * this path and cannot be built here, its two README screenshots are what the oracle is checked against (oracle.h);
 * below their resolution parity is unpinned.  This file restates the reference's algorithm and every function cites the
 * reference lines it follows.  Paths are relative to /root/reference.
 *
 * Arithmetic contract (shared, by specification, with the HIP kernels): IEEE fp32, evaluated exactly as the
 * expressions are written (-ffp-contract=off), with FMA only where fmaf() is written explicitly (dot3, cross,
 * the slab test).  Transcendentals come from one of two tables: glibc (ORC_MATH_LIBM, the literal restatement) or
 * the portable polynomial versions below (ORC_MATH_DET, bit-reproducible on the GPU).
 *
 * Documented deviations from the reference (none changes a valid image beyond rounding):
 *  - env-map lookup clamps the texel index to [0,W-1]x[0,H-1] and acos's argument to [-1,1]
 *    (the reference reads out of bounds / takes acos(1+ulp)=NaN there, CUDATracerProg.cu:813-838);
 *  - `ti` is initialised (CUDATracerProg.cu:211 leaves it uninitialised on a miss; harmless);
 *  - PTX vmin/vmax int tricks (CUDATracerProg.cu:77-87) are restated as fminf/fmaxf (same decisions).
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------------------
// vec3 (CUDATracer/Math/Vec.hpp:306-348, Math/vec/functors.hpp:143-199)
// ------------------------------------------------------------------------------------------------------------
struct v3 { float x, y, z; };
inline v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
inline v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline v3 operator*(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline v3 operator/(v3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
inline float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Vec.hpp:317-320 (a.x*b.x + a.y*b.y + a.z*b.z); contraction pinned to an fma chain
inline float dot3(v3 a, v3 b) { return ffma(a.z, b.z, ffma(a.y, b.y, a.x * b.x)); }
// Vec.hpp:308-313
inline v3 cross3(v3 a, v3 b) {
    return mk(ffma(a.y, b.z, -(b.y * a.z)), ffma(a.z, b.x, -(b.z * a.x)), ffma(a.x, b.y, -(b.x * a.y)));
}
inline float length3(v3 a) { return sqrtf(dot3(a, a)); }                  // Vec.hpp:340-343
inline v3 normalize3(v3 a) { return (a * 1.f) / sqrtf(dot3(a, a)); }      // Vec.hpp:333-336: (v*1.f)/sqrt(dot)
inline v3 reflect3(v3 i, v3 n) { return i - n * dot3(n, i) * 2.0f; }      // Vec.hpp:346-348
inline float saturatef(float x) { return !(x > 0.f) ? 0.f : (x > 1.f ? 1.f : x); } // __saturatef: NaN -> 0
inline v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// host-side (MSVC, no contraction) flavours used by scene construction only
inline float hdot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 hcross3(v3 a, v3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline v3 hnormalize3(v3 a) { return (a * 1.f) / sqrtf(hdot3(a, a)); }

// ------------------------------------------------------------------------------------------------------------
// RNG (CUDATracerProg.cu:28-76)
// ------------------------------------------------------------------------------------------------------------
inline uint32_t tea16(uint32_t val0, uint32_t val1) {           // LCG<16>::init, :40-52
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (unsigned n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
struct Lcg {
    uint32_t state;
    inline uint32_t NextInt() { state = 1664525u * state + 1013904223u; return state; }   // :55-60
    inline float NextFloat() { return (NextInt() & 0x00FFFFFFu) / (float)0x01000000; }    // :62-73
};
// Node-keyed stream (build-side, SURVEY section 7 "RNG parity"): the per-sample TEA state hashed with the ray's node id
// in the binary recursion tree (root 1, specular child 2k, second child 2k+1).  lowbias32 finaliser.
inline uint32_t node_seed(uint32_t s0, uint32_t node) {
    uint32_t x = s0 ^ (node * 0x9E3779B9u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// ------------------------------------------------------------------------------------------------------------
// Deterministic transcendentals (Cephes-style single precision kernels; only + - * / sqrt fma and bit ops)
// ------------------------------------------------------------------------------------------------------------
inline void det_sincosf(float x, float* s, float* c) {
    // valid for |x| < ~1e4; used on [0, 2pi]
    float kf = floorf(ffma(x, 0.636619772367581343f, 0.5f));
    int k = (int)kf;
    float r = ffma(-kf, 1.5703125f, x);
    r = ffma(-kf, 4.837512969970703125e-4f, r);
    r = ffma(-kf, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = ffma(ffma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = ffma(ps * z, r, r);
    float pc = ffma(ffma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = ffma(pc, z * z, ffma(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s = sn;  *c = cs;  break;
        case 1: *s = cs;  *c = -sn; break;
        case 2: *s = -sn; *c = -cs; break;
        default: *s = -cs; *c = sn; break;
    }
}
inline float det_atan_pos(float t) {   // t >= 0
    float y0 = 0.f;
    if (t > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (t - 1.0f) / (t + 1.0f); }
    float z = t * t;
    float p = ffma(ffma(ffma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + ffma(p * z, t, t);
}
inline float det_atan2f(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax == 0.f) a = (ay == 0.f) ? 0.f : 1.5707963267948966f;
    else a = det_atan_pos(ay / ax);
    if (x < 0.f) a = 3.14159265358979323846f - a;
    return (y < 0.f) ? -a : a;
}
inline float det_asin_poly(float x) {  // |x| <= 0.5
    float z = x * x;
    float p = ffma(ffma(ffma(ffma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                   1.6666752422e-1f);
    return ffma(p * z, x, x);
}
inline float det_acosf(float x) {      // x in [-1,1]
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asin_poly(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asin_poly(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - det_asin_poly(x);
}
inline float det_logf(float x) {       // x > 0, normal
    uint32_t bits; memcpy(&bits, &x, 4);
    int e = (int)((bits >> 23) & 0xff) - 126;                 // x = m * 2^e, m in [0.5,1)
    bits = (bits & 0x007fffffu) | 0x3f000000u;
    float m; memcpy(&m, &bits, 4);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = ffma(p, m, -1.1514610310e-1f); p = ffma(p, m, 1.1676998740e-1f); p = ffma(p, m, -1.2420140846e-1f);
    p = ffma(p, m, 1.4249322787e-1f);  p = ffma(p, m, -1.6668057665e-1f); p = ffma(p, m, 2.0000714765e-1f);
    p = ffma(p, m, -2.4999993993e-1f); p = ffma(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = ffma(-2.12194440e-4f, fe, y);
    y = ffma(-0.5f, z, y);
    float r = m + y;
    return ffma(0.693359375f, fe, r);
}
inline float det_expf(float x) {       // |x| < 80
    float nf = floorf(ffma(x, 1.44269504088896341f, 0.5f));
    int n = (int)nf;
    float r = ffma(-nf, 0.693359375f, x);
    r = ffma(nf, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = ffma(p, r, 1.3981999507e-3f); p = ffma(p, r, 8.3334519073e-3f); p = ffma(p, r, 4.1665795894e-2f);
    p = ffma(p, r, 1.6666665459e-1f); p = ffma(p, r, 5.0000001201e-1f);
    float y = ffma(p, z, r) + 1.0f;
    uint32_t sb = (uint32_t)(n + 127) << 23;                   // 2^n, n in [-126,127]
    float sc; memcpy(&sc, &sb, 4);
    return y * sc;
}
inline float det_pow_gamma(float x) {  // x^(1/2.2), x in [0,1]
    if (!(x > 1e-30f)) return 0.f;
    return det_expf((1.0f / 2.2f) * det_logf(x));
}
inline float pow5_mul(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
inline float pow10_mul(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; return x8 * x2; }

struct MathFns {
    void (*sincos)(float, float*, float*);
    float (*sin_)(float);
    float (*atan2_)(float, float);
    float (*acos_)(float);
    float (*pow_gamma)(float);
    float (*pow5)(float);
    float (*pow10)(float);
};
void libm_sincos(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
float libm_sin(float x) { return sinf(x); }
float libm_atan2(float y, float x) { return atan2f(y, x); }
float libm_acos(float x) { return acosf(x); }
float libm_pow_gamma(float x) { return powf(x, 1.0f / 2.2f); }
float libm_pow5(float x) { return powf(x, 5); }
float libm_pow10(float x) { return powf(x, 10); }
float det_sin1(float x) { float s, c; det_sincosf(x, &s, &c); return s; }
const MathFns kLibm = { libm_sincos, libm_sin, libm_atan2, libm_acos, libm_pow_gamma, libm_pow5, libm_pow10 };
const MathFns kDet = { det_sincosf, det_sin1, det_atan2f, det_acosf, det_pow_gamma, pow5_mul, pow10_mul };

// ------------------------------------------------------------------------------------------------------------
// Matrices / quaternion / procedural meshes (Math/Matrix.hpp:116-146,288-346; Math/Quaternion.hpp:79-81;
// Types.hpp:435-574).  Host code in the reference: plain mul/add, no fma.
// ------------------------------------------------------------------------------------------------------------
struct M4 { float d[16]; };
M4 m4_identity() { M4 m; memset(&m, 0, sizeof m); m.d[0] = m.d[5] = m.d[10] = m.d[15] = 1; return m; }
M4 m4_mul(const M4& A, const M4& B) {                        // Matrix.hpp:131-146
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float u = 0; for (int t = 0; t < 4; t++) u += A.d[i * 4 + t] * B.d[t * 4 + j];
        r.d[i * 4 + j] = u;
    }
    return r;
}
M4 m4_translate(v3 t) { M4 m = m4_identity(); m.d[3] = t.x; m.d[7] = t.y; m.d[11] = t.z; return m; }   // :288-296
M4 m4_scale(v3 s) { M4 m = m4_identity(); m.d[0] = s.x; m.d[5] = s.y; m.d[10] = s.z; return m; }       // :298-306
struct Quat { float r, i, j, k; };
Quat quat_rotate(v3 u, float r) {                              // Quaternion.hpp:79-81
    v3 n = hnormalize3(u);
    float s = sinf(0.5f * r);
    Quat q; q.r = cosf(0.5f * r); q.i = s * n.x; q.j = s * n.y; q.k = s * n.z; return q;
}
M4 m4_rotation(Quat q) {                                       // Matrix.hpp:308-346
    M4 m; memset(&m, 0, sizeof m);
    float qxx = q.i * q.i, qyy = q.j * q.j, qzz = q.k * q.k, qxz = q.i * q.k, qxy = q.i * q.j, qyz = q.j * q.k,
          qwx = q.r * q.i, qwy = q.r * q.j, qwz = q.r * q.k;
    m.d[0] = 1.f - 2.f * (qyy + qzz); m.d[4] = 2.f * (qxy + qwz);       m.d[8] = 2.f * (qxz - qwy);
    m.d[1] = 2.f * (qxy - qwz);       m.d[5] = 1.f - 2.f * (qxx + qzz); m.d[9] = 2.f * (qyz + qwx);
    m.d[2] = 2.f * (qxz + qwy);       m.d[6] = 2.f * (qyz - qwx);       m.d[10] = 1.f - 2.f * (qxx + qyy);
    m.d[15] = 1.f;
    return m;
}
v3 m4_apply(const M4& M, v3 v, float w) {                      // Matrix.hpp:116-129
    return mk(M.d[0] * v.x + M.d[1] * v.y + M.d[2] * v.z + M.d[3] * w,
              M.d[4] * v.x + M.d[5] * v.y + M.d[6] * v.z + M.d[7] * w,
              M.d[8] * v.x + M.d[9] * v.y + M.d[10] * v.z + M.d[11] * w);
}

struct MeshData { std::vector<orc_vertex> v; std::vector<uint32_t> idx; orc_material mat; };
orc_vertex mkvert(v3 p, v3 n, float u, float v) {
    orc_vertex r; r.position[0] = p.x; r.position[1] = p.y; r.position[2] = p.z;
    r.normal[0] = n.x; r.normal[1] = n.y; r.normal[2] = n.z; r.uv[0] = u; r.uv[1] = v; return r;
}
void push_wind(std::vector<uint32_t>& e, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint32_t w[6] = { a, b, c, a, c, d }; e.insert(e.end(), w, w + 6);
}
MeshData make_quad() {                                         // Types.hpp:540-557, size {1,1}, offset 0
    MeshData m; v3 n = mk(0, 1, 0);
    m.v = { mkvert(mk(0, 0, 0), n, 0, 0), mkvert(mk(0, 0, 1), n, 0, 1), mkvert(mk(1, 0, 1), n, 1, 1), mkvert(mk(1, 0, 0), n, 1, 0) };
    push_wind(m.idx, 0, 1, 2, 3);
    return m;
}
MeshData make_cube() {                                         // Types.hpp:435-470
    MeshData m;
    auto quad = [&](int ax, float val, v3 n) {
        const float ab[4][2] = { {0, 0}, {1, 0}, {1, 1}, {0, 1} };
        for (int i = 0; i < 4; i++) {
            float a = ab[i][0], b = ab[i][1]; v3 p;
            if (ax == 0) p = mk(val, a, b); else if (ax == 1) p = mk(a, val, b); else p = mk(a, b, val);
            m.v.push_back(mkvert(p, n, a, b));
        }
    };
    quad(1, 0, mk(0, -1, 0)); quad(1, 1, mk(0, 1, 0));
    quad(0, 1, mk(1, 0, 0));  quad(0, 0, mk(-1, 0, 0));
    quad(2, 1, mk(0, 0, 1));  quad(2, 0, mk(0, 0, -1));
    push_wind(m.idx, 0, 1, 2, 3);     push_wind(m.idx, 7, 6, 5, 4);
    push_wind(m.idx, 8, 9, 10, 11);   push_wind(m.idx, 15, 14, 13, 12);
    push_wind(m.idx, 16, 17, 18, 19); push_wind(m.idx, 23, 22, 21, 20);
    return m;
}
MeshData make_sphere() {                                       // Types.hpp:473-538
    MeshData m; const unsigned seg_y = 40, seg_x = 80;
    m.v.push_back(mkvert(mk(0, 1, 0), mk(0, 1, 0), 0.5f, 1));
    m.v.push_back(mkvert(mk(0, -1, 0), mk(0, -1, 0), 0.5f, 0));
    float delta_x = (float)(M_PI * 2 / seg_x), delta_y = (float)(M_PI / seg_y);
    for (int y = 1; y < (int)seg_y; y++) {
        float zenith = y * delta_y, r = sinf(zenith), height = -cosf(zenith);
        for (int x = 0; x < (int)seg_x; x++) {
            float az = x * delta_x, cx = cosf(az) * r, cz = sinf(az) * r;
            m.v.push_back(mkvert(mk(cx, height, cz), mk(cx, height, cz), 0, 0));  // uv = int/int = 0 (:495,500)
        }
    }
    for (int y = 0; y < (int)seg_y - 2; y++) for (int x = 0; x < (int)seg_x; x++) {
        int nx = (x + 1) % seg_x, ny = y + 1;
        uint32_t l1 = 2 + y * seg_x + x, t1 = 2 + ny * seg_x + x, l2 = 2 + y * seg_x + nx, t2 = 2 + ny * seg_x + nx;
        push_wind(m.idx, l2, l1, t1, t2);
    }
    uint32_t top = (uint32_t)m.v.size() - seg_x, bot = 2;
    for (int x = 0; x < (int)seg_x; x++) {
        uint32_t nx = (x + 1) % seg_x;
        uint32_t w[6] = { 0, top + nx, top + (uint32_t)x, 1, bot + (uint32_t)x, bot + nx }; m.idx.insert(m.idx.end(), w, w + 6);
    }
    return m;
}
MeshData construct_mesh(MeshData m, const orc_material& mat, const M4& T) {   // Types.hpp:560-574
    for (auto& v : m.v) {
        v3 p = m4_apply(T, ld3(v.position), 1.f), n = m4_apply(T, ld3(v.normal), 0.f);
        v.position[0] = p.x; v.position[1] = p.y; v.position[2] = p.z;
        v.normal[0] = n.x; v.normal[1] = n.y; v.normal[2] = n.z;
    }
    m.mat = mat; return m;
}
orc_material zero_mat() { orc_material m; memset(&m, 0, sizeof m); return m; }

std::vector<MeshData> cornell_scene() {                        // CornellBox/tracer.cpp:16-143
    orc_material diff = zero_mat();
    diff.color[0] = diff.color[1] = diff.color[2] = 1.0f; diff.roughness = 0.5f; diff.metallic = 0; diff.opacity = 1; diff.n = 2.5f;
    M4 wall = m4_mul(m4_translate(mk(-2, -2, -2)), m4_scale(mk(4, 1, 4)));
    const float PI = (float)M_PI;
    std::vector<MeshData> s;
    s.push_back(construct_mesh(make_quad(), diff, wall));                                                   // floor
    s.push_back(construct_mesh(make_quad(), diff, m4_mul(m4_rotation(quat_rotate(mk(0, 0, 1), PI)), wall)));             // ceiling
    s.push_back(construct_mesh(make_quad(), diff, m4_mul(m4_rotation(quat_rotate(mk(0, 0, 1), (float)(M_PI / 2))), wall))); // back
    orc_material left = diff;  left.color[0] = 1; left.color[1] = 0; left.color[2] = 0;
    s.push_back(construct_mesh(make_quad(), left, m4_mul(m4_rotation(quat_rotate(mk(1, 0, 0), (float)(M_PI / 2))), wall)));
    orc_material right = diff; right.color[0] = 0; right.color[1] = 1; right.color[2] = 0;
    s.push_back(construct_mesh(make_quad(), right, m4_mul(m4_rotation(quat_rotate(mk(1, 0, 0), (float)(-M_PI / 2))), wall)));
    orc_material light = diff; light.emissiveColor[0] = 1.f; light.emissiveColor[1] = .8f; light.emissiveColor[2] = 0.5f; light.emissiveStrength = 10;
    s.push_back(construct_mesh(make_quad(), light, m4_mul(m4_rotation(quat_rotate(mk(0, 0, 1), PI)), m4_translate(mk(-0.5f, -1.99999f, -0.5f)))));
    orc_material box1 = diff; box1.roughness = 0; box1.metallic = 1;
    s.push_back(construct_mesh(make_cube(), box1, m4_mul(m4_translate(mk(0.5f - 0.5f, -2, -0.5f - 0.5f)),
                 m4_mul(m4_rotation(quat_rotate(mk(0, 1, 0), 0.4f)), m4_scale(mk(1, 2.5f, 1))))));
    orc_material box2 = diff; box2.roughness = 0.5f; box2.metallic = 0;
    s.push_back(construct_mesh(make_cube(), box2, m4_mul(m4_translate(mk(-0.5f - 0.5f, -2, 0.5f - 0.5f)),
                 m4_rotation(quat_rotate(mk(0, 1, 0), -0.4f)))));
    return s;
}
std::vector<MeshData> default_scene() {                        // DefaultScene/tracer.cpp:28-123
    auto setm = [](float r, float g, float b, float rough, float metal, float opac, float n) {
        orc_material m = zero_mat(); m.color[0] = r; m.color[1] = g; m.color[2] = b; m.roughness = rough; m.metallic = metal; m.opacity = opac; m.n = n; return m; };
    orc_material mat = setm(0.9f, 0.7f, 0.1f, 0.01f, 0.95f, 1.f, 1.5f);
    orc_material mat2 = setm(0.5f, 0.5f, 0.9f, 0.0f, 0.1f, 0.2f, 2.7f);
    orc_material mat3 = setm(0.3f, 0.95f, 0.4f, 0.00009f, 0.01f, 0.2f, 1.7f);
    orc_material mat4 = setm(0.1f, 0.95f, 0.84f, 0.7f, 0.9f, 0.3f, 1.5f);
    orc_material ground = setm(0.9f, 0.9f, 0.8f, 0.5f, 0.01f, 1.f, 2.7f);
    M4 cube2 = m4_mul(m4_translate(mk(1.3f, 0, 1.7f)), m4_mul(m4_rotation(quat_rotate(mk(0, 1, 0), 0.2f)), m4_scale(mk(1.2f, 1.2f, 1.2f))));
    M4 cube1 = m4_translate(mk(0, 0.1f, -0.5f));
    M4 sph1 = m4_mul(m4_translate(mk(-1.5f, 0.5f, 1.2f)), m4_scale(mk(.5f, .5f, .5f)));
    M4 sph2 = m4_mul(m4_translate(mk(-1.2f, 1, -1.2f)), m4_scale(mk(.8f, .8f, .8f)));
    M4 quad1 = m4_mul(m4_translate(mk(-4, 0, -4)), m4_scale(mk(8.f, 1.f, 8.f)));
    std::vector<MeshData> s;
    s.push_back(construct_mesh(make_cube(), mat, cube1));
    s.push_back(construct_mesh(make_cube(), mat2, cube2));
    s.push_back(construct_mesh(make_quad(), ground, quad1));
    s.push_back(construct_mesh(make_sphere(), mat3, sph1));
    s.push_back(construct_mesh(make_sphere(), mat4, sph2));
    return s;
}
const std::vector<MeshData>& builtin(int which) {
    static std::vector<MeshData> c = cornell_scene(), d = default_scene();
    return which == 0 ? c : d;
}

// ------------------------------------------------------------------------------------------------------------
// Reference median-split BVH2, one per mesh (Types.hpp:312-413)
// ------------------------------------------------------------------------------------------------------------
struct Box { v3 mn, mx; };
struct RefNode { Box box; bool leaf; int l, r; std::vector<uint32_t> tris; };   // tris = local triangle ids (leaf)
struct OMesh {
    std::vector<orc_vertex> v; std::vector<uint32_t> idx; orc_material mat;
    std::vector<RefNode> nodes; int root; uint32_t triBase;
};
int build_median(OMesh& m, std::vector<uint32_t>& tri, const std::vector<v3>& centers, int ttl) {
    Box box; box.mn = mk(INFINITY, INFINITY, INFINITY); box.mx = mk(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t t : tri) for (int k = 0; k < 3; k++) {
        v3 p = ld3(m.v[m.idx[t * 3 + k]].position);
        if (p.x < box.mn.x) box.mn.x = p.x; if (p.x > box.mx.x) box.mx.x = p.x;
        if (p.y < box.mn.y) box.mn.y = p.y; if (p.y > box.mx.y) box.mx.y = p.y;
        if (p.z < box.mn.z) box.mn.z = p.z; if (p.z > box.mx.z) box.mx.z = p.z;
    }
    RefNode n; n.box = box; n.l = n.r = -1;
    if (ttl <= 0 || tri.size() <= 4u) {                        // BVH_MIN_TRICNT 4, :345
        n.leaf = true; n.tris = tri;
        m.nodes.push_back(n); return (int)m.nodes.size() - 1;
    }
    n.leaf = false;
    v3 cmn = mk(INFINITY, INFINITY, INFINITY), cmx = mk(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t t : tri) { v3 c = centers[t];
        cmn.x = std::min(cmn.x, c.x); cmx.x = std::max(cmx.x, c.x); cmn.y = std::min(cmn.y, c.y); cmx.y = std::max(cmx.y, c.y);
        cmn.z = std::min(cmn.z, c.z); cmx.z = std::max(cmx.z, c.z); }
    float lx = cmx.x - cmn.x, ly = cmx.y - cmn.y, lz = cmx.z - cmn.z;
    int ax; if (lx >= ly) ax = (lx >= lz) ? 0 : 2; else ax = (ly >= lz) ? 1 : 2;     // :379-385
    size_t half = tri.size() / 2;
    std::nth_element(tri.begin(), tri.begin() + half, tri.end(), [&](uint32_t a, uint32_t b) {
        const float* ca = &centers[a].x; const float* cb = &centers[b].x; return ca[ax] < cb[ax]; });
    std::vector<uint32_t> lt(tri.begin(), tri.begin() + half), rt(tri.begin() + half, tri.end());
    int self = (int)m.nodes.size(); m.nodes.push_back(n);
    int l = build_median(m, lt, centers, ttl - 1);
    int r = build_median(m, rt, centers, ttl - 1);
    m.nodes[self].l = l; m.nodes[self].r = r;
    return self;
}
void build_mesh_bvh(OMesh& m) {
    uint32_t nt = (uint32_t)m.idx.size() / 3;
    std::vector<uint32_t> tri(nt); std::vector<v3> centers(nt);
    for (uint32_t t = 0; t < nt; t++) {
        tri[t] = t;
        v3 c = ld3(m.v[m.idx[t * 3]].position) + ld3(m.v[m.idx[t * 3 + 1]].position) + ld3(m.v[m.idx[t * 3 + 2]].position);
        centers[t] = c / 3.0f;                                  // :326-327
    }
    m.nodes.clear(); m.root = build_median(m, tri, centers, 32);   // BVH_DEPTH 32
}

} // namespace

// ------------------------------------------------------------------------------------------------------------
// Scene object
// ------------------------------------------------------------------------------------------------------------
struct orc_scene {
    std::vector<OMesh> meshes;
    orc_skylight sky;
    std::vector<float> env; uint32_t envW = 0, envH = 0;
    // global triangle table (mesh order, then triangle order)
    std::vector<uint32_t> triMesh, triLocal;
    // product's flat BVH
    std::vector<float> fnodes, ftris;
    uint32_t fwidth = 2;             // 2: BVH2 nodes, 4: quantised 4-wide nodes
    int estimator = 0;
    bool anyhitLast = false, anyEmits = false;     // see LaunchRayRecur
    uint32_t emitRoot = 0, nRefsMain = 0;          // the product's "emitters first" rule (orc_scene_set_emitter_tree); 0 = off
    std::vector<float> envCdf;       // rowCdf[H+1], then H x colCdf[W+1]
    float neeScale = 0.f;            // pi / (0.98 * W * H): p_cap * texel solid angle / sin(theta)
    // emissive triangles (estimator 1): 15 floats each [v0 e1 e2 n Le], area CDF [n+1], total area / (2 pi 0.98)
    std::vector<float> emitTri, emitCdf; float emitScale = 0.f;
};

namespace {

struct Hit { float t, u, v; int tri; bool front; };   // tri = global id
struct alignas(128) Counters { uint64_t rays = 0, nodes = 0, tris = 0, hits = 0, shadow = 0, anyhit = 0, emitfirst = 0; };   // one cache line per thread

// Moller-Trumbore on (v0, e1, e2) -- CUDATracerProg.cu:232-263.  Accepts iff t>0 && t<tmin (strict), or -- when
// `id`/`bestId` are given (>= 0) -- on an exact tie t==tmin if the candidate's global triangle id is lower.  The
// reference lets the first triangle FOUND win a tie (leaf order, then mesh order: SURVEY F8); across meshes that is
// the lower mesh index, i.e. the lower global id.  Breaking ties by id makes the closest hit independent of the BVH.
inline bool mt_test(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float tmin, float* t_, float* u_, float* v_, bool* front,
                    int id = -1, int bestId = -1) {
    v3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (fabsf(det) < FLT_EPSILON) return false;
    float invDet = 1 / det;
    v3 tvec = o - v0;
    float u = dot3(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    v3 qvec = cross3(tvec, e1);
    float v = dot3(d, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    float t = dot3(e2, qvec) * invDet;
    if (t > 0 && (t < tmin || (t == tmin && id >= 0 && id < bestId))) { *t_ = t; *u_ = u; *v_ = v; *front = det > 0; return true; }
    return false;
}
inline void tri_of(const OMesh& m, uint32_t t, v3* v0, v3* e1, v3* e2) {
    v3 a = ld3(m.v[m.idx[t * 3]].position), b = ld3(m.v[m.idx[t * 3 + 1]].position), c = ld3(m.v[m.idx[t * 3 + 2]].position);
    *v0 = a; *e1 = b - a; *e2 = c - a;
}
struct RayPre { v3 o, d, idir, ood; };
inline RayPre ray_pre(v3 o, v3 d) {                             // CUDATracerProg.cu:567-574
    const float ooeps = 8.271806125530277e-25f;                 // exp2f(-80)
    RayPre r; r.o = o; r.d = d;
    r.idir.x = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
    r.idir.y = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
    r.idir.z = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
    r.ood = mk(o.x * r.idir.x, o.y * r.idir.y, o.z * r.idir.z);
    return r;
}
// min/max without libm calls (gcc cannot inline fminf/fmaxf without -ffinite-math-only); operands are never NaN here
inline float fmin_(float a, float b) { return a < b ? a : b; }
inline float fmax_(float a, float b) { return a > b ? a : b; }
// slab test, CUDATracerProg.cu:651-658 with spanBegin/EndKepler (:86-87): tnear = max(0,...), tfar = min(hitT,...)
inline bool slab(const RayPre& r, const float* mn, const float* mx, float hitT, float* tnear) {
    float lox = ffma(mn[0], r.idir.x, -r.ood.x), hix = ffma(mx[0], r.idir.x, -r.ood.x);
    float loy = ffma(mn[1], r.idir.y, -r.ood.y), hiy = ffma(mx[1], r.idir.y, -r.ood.y);
    float loz = ffma(mn[2], r.idir.z, -r.ood.z), hiz = ffma(mx[2], r.idir.z, -r.ood.z);
    float tmin = fmax_(fmax_(fmin_(lox, hix), fmin_(loy, hiy)), fmax_(fmin_(loz, hiz), 0.0f));
    float tmax = fmin_(fmin_(fmax_(lox, hix), fmax_(loy, hiy)), fmin_(fmax_(loz, hiz), hitT));
    *tnear = tmin;
    return tmin <= tmax;
}

// --- reference semantics: per mesh, no t-culling, near child first, leaves as found (CUDATracerProg.cu:584-791) ---
void ref_visit(const orc_scene& s, const OMesh& m, int ni, const RayPre& r, Hit* best, Counters* c) {
    const RefNode& n = m.nodes[ni];
    if (n.leaf) {
        // NearestIntersectSubset (:185-303): local closest over the leaf, then strict compare with the global tmin (:760)
        float tmin = INFINITY; Hit h; h.tri = -1; h.t = INFINITY; h.u = h.v = 0; h.front = false;
        for (uint32_t t : n.tris) {
            v3 v0, e1, e2; tri_of(m, t, &v0, &e1, &e2); c->tris++;
            float tt, uu, vv; bool ff;
            if (mt_test(r.o, r.d, v0, e1, e2, tmin, &tt, &uu, &vv, &ff)) { tmin = tt; h.t = tt; h.u = uu; h.v = vv; h.front = ff; h.tri = (int)(m.triBase + t); }
        }
        if (h.tri >= 0 && h.t < best->t) *best = h;
        return;
    }
    c->nodes++;
    const RefNode& a = m.nodes[n.l]; const RefNode& b = m.nodes[n.r];
    float t0, t1;
    bool h0 = slab(r, &a.box.mn.x, &a.box.mx.x, INFINITY, &t0);   // hitT stays +inf (:595,658)
    bool h1 = slab(r, &b.box.mn.x, &b.box.mx.x, INFINITY, &t1);
    if (h0 && h1) {
        if (t1 < t0) { ref_visit(s, m, n.r, r, best, c); ref_visit(s, m, n.l, r, best, c); }
        else { ref_visit(s, m, n.l, r, best, c); ref_visit(s, m, n.r, r, best, c); }
    } else if (h0) ref_visit(s, m, n.l, r, best, c);
    else if (h1) ref_visit(s, m, n.r, r, best, c);
}
Hit isect_ref(const orc_scene& s, v3 o, v3 d, Counters* c) {
    RayPre r = ray_pre(o, d);
    Hit best; best.t = INFINITY; best.tri = -1; best.u = best.v = 0; best.front = false;
    for (const OMesh& m : s.meshes) ref_visit(s, m, m.root, r, &best, c);   // root box is never tested (:593-598)
    return best;
}
Hit isect_brute(const orc_scene& s, v3 o, v3 d, Counters* c) {
    Hit best; best.t = INFINITY; best.tri = -1; best.u = best.v = 0; best.front = false;
    for (const OMesh& m : s.meshes) {
        uint32_t nt = (uint32_t)m.idx.size() / 3;
        for (uint32_t t = 0; t < nt; t++) {
            v3 v0, e1, e2; tri_of(m, t, &v0, &e1, &e2); c->tris++;
            float tt, uu, vv; bool ff;
            if (mt_test(o, d, v0, e1, e2, best.t, &tt, &uu, &vv, &ff, (int)(m.triBase + t), best.tri)) { best.t = tt; best.u = uu; best.v = vv; best.front = ff; best.tri = (int)(m.triBase + t); }
        }
    }
    return best;
}
// --- product flat BVH (layout: include/mipt.h).  Child ref: bit31 = leaf, bits 27..30 = count, bits 0..26 = first ref;
//     inner = node index.  Order: nearer child first, farther pushed; leaves tested when they become current; hitT
//     shrinks as hits are found and culls later box tests (tnear <= hitT).  No re-test at pop. ---
inline int32_t fbits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
// How a ray ends early in the product's wave-stack integrator (kernels.hip, phase D; work-counting options, same image):
//   kClosest     -- never: the closest hit
//   kAnyhit      -- at the end of the first leaf that left it with a hit (last-level ray, no mesh of the scene emits)
//   kEmitFirst   -- last-level ray in a scene with emitters: it starts in the emitters' tree with the main root waiting on the
//                   stack; at the end of a leaf it is over if its closest hit so far does not emit, or -- under a black sky, no
//                   hit yet -- if the entry it is about to pop is the main root or the leaf belonged to the main tree
enum FlatEnd { kClosest = 0, kAnyhit = 1, kEmitFirst = 2 };
inline bool sky_black(const orc_scene& s) { return s.env.empty() && s.sky.intensity == 0.f; }
Hit isect_flat(const orc_scene& s, v3 o, v3 d, Counters* c, int end = kClosest) {
    RayPre r = ray_pre(o, d);
    Hit best; best.t = INFINITY; best.tri = -1; best.u = best.v = 0; best.front = false;
    if (s.fnodes.empty()) return best;
    uint32_t stack[128]; int sp = 0;
    uint32_t cur = 0;
    bool bestEmits = false; const bool black = sky_black(s);
    if (end == kEmitFirst) { stack[sp++] = 0u; cur = s.emitRoot; }
    while (true) {
        if (cur & 0x80000000u) {
            uint32_t cnt = (cur >> 27) & 0xF, first = cur & 0x07FFFFFFu;
            for (uint32_t i = first; i < first + cnt; i++) {
                const float* T = &s.ftris[(size_t)i * 12]; c->tris++;
                float tt, uu, vv; bool ff;
                if (mt_test(o, d, ld3(T), ld3(T + 4), ld3(T + 8), best.t, &tt, &uu, &vv, &ff, fbits(T[3]), best.tri)) {
                    best.t = tt; best.u = uu; best.v = vv; best.front = ff; best.tri = fbits(T[3]); bestEmits = (fbits(T[11]) & 2) != 0;
                }
            }
            if (end == kAnyhit && best.tri >= 0) return best;  // the product's last-level shortcut: the leaf is finished, then the ray ends
            if (end == kEmitFirst && (best.tri >= 0 ? !bestEmits : (black && ((sp > 0 && stack[sp - 1] == 0u) || (cnt != 0u && first < s.nRefsMain))))) return best;
        } else {
            const float* N = &s.fnodes[(size_t)cur * 16]; c->nodes++;
            float t0, t1;
            bool h0 = slab(r, N + 0, N + 3, best.t, &t0);
            bool h1 = slab(r, N + 6, N + 9, best.t, &t1);
            uint32_t c0 = (uint32_t)fbits(N[12]), c1 = (uint32_t)fbits(N[13]);
            if (h0 || h1) {
                uint32_t first = h0 ? c0 : c1;
                if (h0 && h1) {
                    uint32_t second = c1;
                    if (t1 < t0) { first = c1; second = c0; }
                    stack[sp++] = second;
                }
                cur = first;
                continue;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}
// --- product 4-wide quantised layout: node = [origin xyz][scale.z][qlo.x qlo.y qlo.z qhi.x qhi.y qhi.z: one byte
//     per child][scale.x][scale.y][ref0..ref3], scales = powers of two as floats.  The ray is moved into the node's
//     quantised frame exactly as the kernel does (kernels.hip wide_visit): t = q * (idir * scale) + (origin - o) * idir.  Hit children are visited nearest
//     first (5-exchange sorting network, swap iff tb < ta), the others pushed farthest first. ---
inline float ubits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
Hit isect_flat4(const orc_scene& s, v3 o, v3 d, Counters* c, int end = kClosest) {
    RayPre r = ray_pre(o, d);
    Hit best; best.t = INFINITY; best.tri = -1; best.u = best.v = 0; best.front = false;
    if (s.fnodes.empty()) return best;
    uint32_t stack[256]; int sp = 0;
    uint32_t cur = 0;
    bool bestEmits = false; const bool black = sky_black(s);
    if (end == kEmitFirst) { stack[sp++] = 0u; cur = s.emitRoot; }
    while (true) {
        if (cur & 0x80000000u) {
            uint32_t cnt = (cur >> 27) & 0xF, first = cur & 0x07FFFFFFu;
            for (uint32_t i = first; i < first + cnt; i++) {
                const float* T = &s.ftris[(size_t)i * 12]; c->tris++;
                float tt, uu, vv; bool ff;
                if (mt_test(o, d, ld3(T), ld3(T + 4), ld3(T + 8), best.t, &tt, &uu, &vv, &ff, fbits(T[3]), best.tri)) {
                    best.t = tt; best.u = uu; best.v = vv; best.front = ff; best.tri = fbits(T[3]); bestEmits = (fbits(T[11]) & 2) != 0;
                }
            }
            if (end == kAnyhit && best.tri >= 0) return best;  // the product's last-level shortcut: the leaf is finished, then the ray ends
            if (end == kEmitFirst && (best.tri >= 0 ? !bestEmits : (black && ((sp > 0 && stack[sp - 1] == 0u) || (cnt != 0u && first < s.nRefsMain))))) return best;
        } else {
            const float* N = &s.fnodes[(size_t)cur * 16]; c->nodes++;
            const float ix = r.idir.x * N[10], iy = r.idir.y * N[11], iz = r.idir.z * N[3];
            const float ox = (N[0] - r.o.x) * r.idir.x, oy = (N[1] - r.o.y) * r.idir.y, oz = (N[2] - r.o.z) * r.idir.z;
            const uint32_t lx = (uint32_t)fbits(N[4]), ly = (uint32_t)fbits(N[5]), lz = (uint32_t)fbits(N[6]);
            const uint32_t hx = (uint32_t)fbits(N[7]), hy = (uint32_t)fbits(N[8]), hz = (uint32_t)fbits(N[9]);
            float t[4]; uint32_t ref[4]; int n = 0;
            for (int i = 0; i < 4; i++) {
                const int sh = 8 * i;
                ref[i] = (uint32_t)fbits(N[12 + i]);
                float tlx = ffma((float)((lx >> sh) & 0xFFu), ix, ox), thx = ffma((float)((hx >> sh) & 0xFFu), ix, ox);
                float tly = ffma((float)((ly >> sh) & 0xFFu), iy, oy), thy = ffma((float)((hy >> sh) & 0xFFu), iy, oy);
                float tlz = ffma((float)((lz >> sh) & 0xFFu), iz, oz), thz = ffma((float)((hz >> sh) & 0xFFu), iz, oz);
                float tmin = fmax_(fmax_(fmin_(tlx, thx), fmin_(tly, thy)), fmax_(fmin_(tlz, thz), 0.0f));
                float tmax = fmin_(fmin_(fmax_(tlx, thx), fmax_(tly, thy)), fmin_(fmax_(tlz, thz), best.t));
                // empty slots carry an inverted box and a zero-triangle leaf reference.  The kernel picks entry / exit planes by the
                // ray's octant, which an inverted box never passes; min / max per axis, as written here, would turn it the right
                // way round again, so the reference is looked at as well (same decision as the kernel's for every slot)
                const bool hit = tmin <= tmax && ref[i] != 0x80000000u;
                t[i] = hit ? tmin : INFINITY;
                n += hit ? 1 : 0;
            }
            auto ce = [&](int a, int b) { if (t[b] < t[a]) { std::swap(t[a], t[b]); std::swap(ref[a], ref[b]); } };
            ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2);
            if (n > 0) {
                if (n > 3) stack[sp++] = ref[3];
                if (n > 2) stack[sp++] = ref[2];
                if (n > 1) stack[sp++] = ref[1];
                cur = ref[0];
                continue;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}
inline Hit isect(const orc_scene& s, int mode, v3 o, v3 d, Counters* c, int end = kClosest) {
    c->rays++;
    Hit h = mode == ORC_ISECT_REF ? isect_ref(s, o, d, c) : mode == ORC_ISECT_FLAT ? (s.fwidth == 4 ? isect_flat4(s, o, d, c, end) : isect_flat(s, o, d, c, end)) : isect_brute(s, o, d, c);
    if (h.tri >= 0) c->hits++;
    return h;
}

// ------------------------------------------------------------------------------------------------------------
// Sampling (CUDATracerProg.cu:93-180)
// ------------------------------------------------------------------------------------------------------------
inline void cone_basis(v3 dir, v3* right, v3* up) {            // :99-104
    float rx = dir.z, ry = -dir.x;
    float rlen = sqrtf(0.f + rx * rx + ry * ry);               // generic dot<T,N> accumulates from 0 (Vec.hpp:323-329)
    *right = rlen <= 0 ? mk(0, 0, 1) : mk(rx, 0, ry) / rlen;
    *up = cross3(*right, dir);
}
v3 RandDirInsideCone(Lcg& rng, v3 dir, float size, const MathFns& M) {   // :93-135
    v3 right, up; cone_basis(dir, &right, &up);
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(M_PI * 2 * u);
    float sx = 1 - 2 * v * size;
    float sn, cs; M.sincos(theta, &sn, &cs);
    float sy = sqrtf(1 - sx * sx) * cs;
    float sz = sqrtf(1 - sx * sx) * sn;
    return sx * dir + sy * up + sz * right;
}
v3 GGXRandDirInsideCone(Lcg& rng, v3 dir, float size, const MathFns& M) {   // :137-180
    v3 right, up; cone_basis(dir, &right, &up);
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(M_PI * 2 * u);
    float phi = M.atan2_(size * sqrtf(v), sqrtf(1 - v));
    float sx = 1 - M.sin_(phi);                                  // (sic) :160
    float sn, cs; M.sincos(theta, &sn, &cs);
    float sy = sqrtf(1 - sx * sx) * cs;
    float sz = sqrtf(1 - sx * sx) * sn;
    return sx * dir + sy * up + sz * right;
}

// ---- estimator 2 (specification: include/mipt.h): the same two lobes, sampled the textbook way ------------------
// diffuse lobe: cosine-weighted hemisphere about N (Malley), same basis and the same two draws as RandDirInsideCone
v3 CosineDirAbout(Lcg& rng, v3 N, const MathFns& M) {
    v3 right, up; cone_basis(N, &right, &up);
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(M_PI * 2 * u);
    float sn, cs; M.sincos(theta, &sn, &cs);
    float r = sqrtf(v), cz = sqrtf(1 - v);
    return cz * N + (r * cs) * up + (r * sn) * right;
}
// specular lobe: microfacet normal from the GGX distribution of visible normals (Heitz 2018), alpha = roughness,
// seen from V = -d; roughness 0 returns N exactly (as the reference's cone does).  Two draws.
v3 GgxVisibleNormal(Lcg& rng, v3 N, v3 d, float alpha, const MathFns& M) {
    float u1 = rng.NextFloat(), u2 = rng.NextFloat();
    if (!(alpha > 0.f)) return N;
    v3 right, up; cone_basis(N, &right, &up);
    v3 V = -d;
    v3 Ve = mk(dot3(V, up), dot3(V, right), fmaxf(dot3(V, N), 1e-4f));
    v3 Vh = normalize3(mk(alpha * Ve.x, alpha * Ve.y, Ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.f ? mk(-Vh.y, Vh.x, 0.f) / sqrtf(lensq) : mk(1.f, 0.f, 0.f);
    v3 T2 = cross3(Vh, T1);
    float r = sqrtf(u1);
    float phi = (float)(M_PI * 2 * u2);
    float sn, cs; M.sincos(phi, &sn, &cs);
    float t1 = r * cs, t2 = r * sn;
    float sw = 0.5f * (1 + Vh.z);
    t2 = (1 - sw) * sqrtf(1 - t1 * t1) + sw * t2;
    float nz = sqrtf(fmaxf(0.f, 1 - t1 * t1 - t2 * t2));
    v3 Nh = T1 * t1 + T2 * t2 + Vh * nz;
    v3 Ne = normalize3(mk(alpha * Nh.x, alpha * Nh.y, fmaxf(1e-6f, Nh.z)));
    return Ne.x * up + Ne.y * right + Ne.z * N;
}
// Smith masking of the outgoing direction (what VNDF sampling leaves of the BRDF weight besides Fresnel)
inline float GgxG1(float nl, float alpha) {
    if (!(nl > 0.f)) return 0.f;
    float a2 = alpha * alpha;
    return (2 * nl) / (nl + sqrtf(a2 + (1 - a2) * (nl * nl)));
}

// ------------------------------------------------------------------------------------------------------------
// Integrator (CUDATracerProg.cu:367-541, 796-846)
// ------------------------------------------------------------------------------------------------------------
struct Ctx {
    const orc_scene* s; int isect; int rng_mode; const MathFns* M; Counters* c; uint32_t s0; uint32_t pixel_rays;
};

// ---- env-map importance sampling (estimator 1; specification shared with the product, csrc/api.hip + kernels.hip) ----
// Tables, in double, in this order: weight(iv,iu) = ((double)r + g + b) * sin(pi (iv + 0.5) / H); per row the running sum
// divided by the row total (uniform if the row is black); the marginal likewise over row totals; stored as float.
void build_env_cdf(orc_scene& s) {
    const uint32_t W = s.envW, H = s.envH;
    s.envCdf.clear(); s.neeScale = 0.f;
    if (s.env.empty() || !W || !H) return;
    s.envCdf.assign((size_t)(H + 1) + (size_t)H * (W + 1), 0.f);
    float* row = s.envCdf.data(); float* col = row + (H + 1);
    std::vector<double> rowTotal(H), acc(W + 1);
    for (uint32_t iv = 0; iv < H; iv++) {
        const double sinT = sin(M_PI * ((double)iv + 0.5) / (double)H);
        acc[0] = 0.0;
        for (uint32_t iu = 0; iu < W; iu++) {
            const float* p = &s.env[((size_t)iv * W + iu) * 3];
            acc[iu + 1] = acc[iu] + ((double)p[0] + (double)p[1] + (double)p[2]) * sinT;
        }
        rowTotal[iv] = acc[W];
        float* c = col + (size_t)iv * (W + 1);
        for (uint32_t iu = 0; iu <= W; iu++) c[iu] = acc[W] > 0.0 ? (float)(acc[iu] / acc[W]) : (float)((double)iu / (double)W);
        c[W] = 1.0f;
    }
    double total = 0.0; for (uint32_t iv = 0; iv < H; iv++) total += rowTotal[iv];
    double run = 0.0;
    for (uint32_t iv = 0; iv <= H; iv++) {
        row[iv] = total > 0.0 ? (float)(run / total) : (float)((double)iv / (double)H);
        if (iv < H) run += rowTotal[iv];
    }
    row[H] = 1.0f;
    s.neeScale = (float)(M_PI / 0.98 / ((double)W * (double)H));
}
// Emitter table: every triangle (global id order) of a mesh with emissiveStrength > 0 and a non-black emissive colour
// and a non-zero area.  In double from the float vertices: c = cross(e1, e2), area = 0.5 |c|, unit normal = c / |c|;
// CDF = running area / total area, as float.  Le = emissiveColor * emissiveStrength (float).
void build_emitters(orc_scene& s) {
    s.emitTri.clear(); s.emitCdf.clear(); s.emitScale = 0.f;
    std::vector<double> area;
    for (const OMesh& m : s.meshes) {
        const orc_material& mt = m.mat;
        if (!(mt.emissiveStrength > 0.f) || !(mt.emissiveColor[0] > 0.f || mt.emissiveColor[1] > 0.f || mt.emissiveColor[2] > 0.f)) continue;
        const uint32_t nt = (uint32_t)m.idx.size() / 3;
        for (uint32_t t = 0; t < nt; t++) {
            v3 v0, e1, e2; tri_of(m, t, &v0, &e1, &e2);
            const double cx = (double)e1.y * e2.z - (double)e2.y * e1.z, cy = (double)e1.z * e2.x - (double)e2.z * e1.x,
                         cz = (double)e1.x * e2.y - (double)e2.x * e1.y;
            const double len = sqrt(cx * cx + cy * cy + cz * cz);
            if (!(len > 0.0)) continue;
            const float rec[15] = { v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, (float)(cx / len), (float)(cy / len), (float)(cz / len),
                                    mt.emissiveColor[0] * mt.emissiveStrength, mt.emissiveColor[1] * mt.emissiveStrength, mt.emissiveColor[2] * mt.emissiveStrength };
            s.emitTri.insert(s.emitTri.end(), rec, rec + 15);
            area.push_back(0.5 * len);
        }
    }
    if (area.empty()) return;
    double total = 0.0; for (double a : area) total += a;
    s.emitCdf.resize(area.size() + 1);
    double run = 0.0;
    for (size_t i = 0; i <= area.size(); i++) { s.emitCdf[i] = (float)(run / total); if (i < area.size()) run += area[i]; }
    s.emitCdf[area.size()] = 1.0f;
    s.emitScale = (float)(total / (2.0 * M_PI * 0.98));
}
// largest i in [0, n-1] with cdf[i] <= u  (cdf[0] = 0, cdf[n] = 1 > u)
inline uint32_t cdf_find(const float* cdf, uint32_t n, float u) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid; else hi = mid; }
    return lo;
}
// one direction from the table; returns false when the texel has no probability mass.  L = its radiance,
// wgt = p_cap / p_env for a direction inside the diffuse cap.
inline bool sample_env(const orc_scene& s, const MathFns& M, float u1, float u2, v3* dir, v3* L, float* wgt) {
    const uint32_t W = s.envW, H = s.envH;
    const float* row = s.envCdf.data(); 
    const uint32_t iv = cdf_find(row, H, u1);
    const float pr = row[iv + 1] - row[iv];
    const float* col = row + (H + 1) + (size_t)iv * (W + 1);
    const uint32_t iu = cdf_find(col, W, u2);
    const float pc = col[iu + 1] - col[iu];
    if (!(pr > 0.f) || !(pc > 0.f)) return false;
    const float du1 = (u1 - row[iv]) / pr, du2 = (u2 - col[iu]) / pc;
    const float theta = (((float)iv + du1) / (float)H) * 3.14159265358979323846f;
    const float phi = (((float)iu + du2) / (float)W) * 6.28318530717958647692f;     // azimuth + pi
    float st, ct, sp, cp; M.sincos(theta, &st, &ct); M.sincos(phi, &sp, &cp);
    *dir = mk(st * -sp, ct, st * -cp);
    const float* p = &s.env[((size_t)iv * W + iu) * 3];
    *L = mk(p[0], p[1], p[2]);
    *wgt = st * s.neeScale / (pr * pc);
    return true;
}
v3 OnMiss(const Ctx& k, v3 d) {                                // :796-846
    const orc_scene& s = *k.s;
    if (s.env.empty()) {
        float ang = (1 - dot3(d, ld3(s.sky.dir))) / 2;
        return ld3(s.sky.color) * s.sky.intensity * k.M->pow10(ang);
    }
    float dy = fminf(1.0f, fmaxf(-1.0f, d.y));                 // deviation: clamp (see header)
    float zenith = k.M->acos_(dy);
    float azimuth = k.M->atan2_(d.x, d.z);
    float v = (float)(zenith / M_PI);
    float u = (float)(azimuth / (M_PI * 2) + 0.5);
    u *= (float)s.envW; v *= (float)s.envH;
    int u0 = (int)u, v0 = (int)v;
    int uc = std::min(std::max(u0, 0), (int)s.envW - 1), vc = std::min(std::max(v0, 0), (int)s.envH - 1);  // deviation: clamp
    uint32_t a = (uint32_t)vc * s.envW + (uint32_t)uc;
    v3 p00 = mk(s.env[(size_t)a * 3], s.env[(size_t)a * 3 + 1], s.env[(size_t)a * 3 + 2]);
    float du = u - u0, dv = v - v0;                              // the "bilinear" identity arithmetic of :840-843
    v3 p0 = (dv) * p00 + (1 - dv) * p00;
    v3 p1 = (dv) * p00 + (1 - dv) * p00;
    return du * p1 + (1 - du) * p0;
}
// one point on the emissive triangles as seen from x (normal N); false when it adds nothing
inline bool sample_emitter(const orc_scene& s, float r0, float r1, float r2, v3 x, v3 N, float minCos, v3* dir, v3* Lw, float* tmax) {
    const uint32_t n = (uint32_t)s.emitCdf.size() - 1;
    const uint32_t k = cdf_find(s.emitCdf.data(), n, r0);
    const float* E = &s.emitTri[(size_t)k * 15];
    const float su = sqrtf(r1), b1 = 1 - su, b2 = r2 * su;
    v3 y = ld3(E) + ld3(E + 3) * b1 + ld3(E + 6) * b2;
    v3 w = y - x;
    const float dist2 = dot3(w, w), dist = sqrtf(dist2);
    if (!(dist > 0.f)) return false;
    *dir = w / dist;
    if (!(dot3(*dir, N) > minCos)) return false;
    const float cosy = fabsf(dot3(ld3(E + 9), *dir));
    if (!(cosy > 0.f)) return false;
    *Lw = ld3(E + 12) * (cosy * s.emitScale / dist2);
    *tmax = dist * 0.9999f;
    return true;
}
v3 LaunchRayRecur(Ctx& k, v3 o, v3 d, int ttl, Lcg* dfs, uint32_t node, bool noEnv = false, bool noEmit = false) {   // :426-541
    v3 color = mk(0, 0, 0);
    if (ttl <= 0) return color;
    k.pixel_rays++;
    // Work-counting option (orc_scene_set_anyhit_last_level; off by default, FLAT intersector only): the product ends a
    // last-level ray at its first hit when no mesh of the scene emits -- such a ray adds nothing if it hits anything.  The
    // image is the same either way; only node visits / triangle tests (the roofline's algorithmic counts) change.
    const bool anyhit = k.s->anyhitLast && ttl == 1 && !k.s->anyEmits;
    // ... and in a scene with emitters and an emitters' tree (orc_scene_set_emitter_tree; estimators 0 and 2, as in the product)
    // the last-level ray takes the "emitters first" route: what it returns below -- the emission of its hit, or the sky -- is
    // what the closest hit would have given (an emitter that is the closest hit is found and not occluded; any other hit has no
    // emission to add; a miss is a miss either way)
    const bool emitFirst = k.s->anyhitLast && ttl == 1 && k.s->anyEmits && k.s->emitRoot != 0u && k.isect == ORC_ISECT_FLAT && (k.s->estimator == 0 || k.s->estimator == 2);
    if (anyhit) k.c->anyhit++;
    if (emitFirst) k.c->emitfirst++;
    Hit h = isect(*k.s, k.isect, o, d, k.c, anyhit ? kAnyhit : emitFirst ? kEmitFirst : kClosest);
    if (h.tri < 0) return noEnv ? color : OnMiss(k, d);
    if (anyhit) return color;
    const OMesh& m = k.s->meshes[k.s->triMesh[h.tri]];
    uint32_t lt = k.s->triLocal[h.tri];
    const orc_material& mat = m.mat;
    // NearestIntersectSubset tail (:268-294): interpolated, not re-normalised normal; position = o + d*t
    float w0 = 1 - h.u - h.v, w1 = h.u, w2 = h.v;
    v3 n0 = normalize3(ld3(m.v[m.idx[lt * 3]].normal)), n1 = normalize3(ld3(m.v[m.idx[lt * 3 + 1]].normal)),
       n2 = normalize3(ld3(m.v[m.idx[lt * 3 + 2]].normal));
    v3 hn = n0 * w0 + n1 * w1 + n2 * w2;
    v3 P = o + d * h.t;
    v3 N = h.front ? hn : -hn;                                  // :442
    Lcg local; Lcg* rng = dfs;
    if (k.rng_mode == ORC_RNG_NODE) { local.state = node_seed(k.s0, node); rng = &local; }
    const bool pbs = k.s->estimator >= 2;                        // estimators 2, 3: VNDF specular + cosine diffuse
    v3 mN = pbs ? GgxVisibleNormal(*rng, N, d, mat.roughness, *k.M) : GGXRandDirInsideCone(*rng, N, mat.roughness, *k.M);  // :444
    v3 R = reflect3(d, mN);                                      // :446
    v3 c = LaunchRayRecur(k, P + N * 1e-6f, R, ttl - 1, dfs, node * 2);      // :448-458
    c = mk(fmaxf(c.x, 0.f), fmaxf(c.y, 0.f), fmaxf(c.z, 0.f));   // :459 clamp to [0,inf)
    float n1_ = 1.f, n2_ = mat.n; n2_ += FLT_EPSILON;            // :463-466
    if (!h.front) { float t = n1_; n1_ = n2_; n2_ = t; }
    float r0 = (n1_ - n2_) / (n1_ + n2_); r0 *= r0;
    float rr = (1 - dot3(-d, mN));
    rr = k.M->pow5(rr);
    rr = r0 + (1 - r0) * rr;
    rr = saturatef(rr);
    rr = mat.metallic * 1.f + (1 - mat.metallic) * rr;
    float rt = 1 - rr;
    v3 col = ld3(mat.color);
    v3 absorb = mat.metallic * col + (1 - mat.metallic) * mk(1.f, 1.f, 1.f);   // :492
    color = absorb * c * rr;                                     // :493
    if (pbs) color = color * GgxG1(dot3(R, N), mat.roughness);
    v3 o2, d2;
    if (mat.opacity < 1.f) {                                     // :495-516
        v3 front = (d + R) / 2.0f;
        v3 down = d - front;
        float down_len = length3(down);
        down = down_len < 1e-10f ? mk(1, 0, 0) : (down / down_len);
        v3 front_refr = front * (n1_ / n2_);
        float frl = length3(front_refr);
        if (frl > 1) frl = 1;
        float drl = sqrtf(1 - frl * frl);
        d2 = normalize3(front_refr + down * drl);
        o2 = P - N * 1e-6f;
    } else {                                                     // :518-522
        d2 = pbs ? CosineDirAbout(*rng, N, *k.M) : RandDirInsideCone(*rng, N, 0.49f, *k.M);
        o2 = P + N * 1e-6f;
    }
    // estimator 1: what the diffuse child would find by chance -- the environment when it escapes, the emission of the
    // surface it lands on -- is sampled explicitly instead (one occlusion ray each); the child is told to skip those terms
    const bool nee = (k.s->estimator == 1 || k.s->estimator == 3) && k.rng_mode == ORC_RNG_NODE && mat.opacity == 1.f && ttl - 1 > 0;
    const float minCos = pbs ? 0.f : 0.02f;                       // estimator 3: the cosine lobe covers the whole hemisphere
    const bool neeEnv = nee && !k.s->envCdf.empty(), neeEmit = nee && !k.s->emitCdf.empty();
    c = LaunchRayRecur(k, o2, d2, ttl - 1, dfs, node * 2 + 1, neeEnv, neeEmit);    // :525
    if (neeEnv) {
        float u1 = rng->NextFloat(), u2 = rng->NextFloat();
        v3 wdir, L; float wgt;
        if (sample_env(*k.s, *k.M, u1, u2, &wdir, &L, &wgt) && dot3(wdir, N) > minCos) {
            k.pixel_rays++; k.c->shadow++;
            Hit sh = isect(*k.s, k.isect, o2, wdir, k.c);
            if (pbs) wgt = wgt * (dot3(wdir, N) * 1.96f);          // (cos / pi) / p_cap: the lobe's density instead of the cap's
            if (sh.tri < 0) c = c + L * wgt;
        }
    }
    if (neeEmit) {
        float r0 = rng->NextFloat(), r1 = rng->NextFloat(), r2 = rng->NextFloat();
        v3 wdir, Lw; float tmax;
        if (sample_emitter(*k.s, r0, r1, r2, o2, N, minCos, &wdir, &Lw, &tmax)) {
            k.pixel_rays++; k.c->shadow++;
            Hit sh = isect(*k.s, k.isect, o2, wdir, k.c);
            if (pbs) Lw = Lw * (dot3(wdir, N) * 1.96f);
            if (!(sh.tri >= 0 && sh.t < tmax)) c = c + Lw;
        }
    }
    color = color + col * c * rt * ((mat.opacity == 1.f) ? 1.f : (1 - mat.opacity));   // :526
    if (!noEmit) color = color + ld3(mat.emissiveColor) * mat.emissiveStrength;        // :528
    return color;
}
struct CamPre { float pixelSize, origin_x, origin_y; };
CamPre cam_pre(const orc_settings& s) {                          // :369-397
    float num_x = (float)s.viewportWidth, num_y = (float)s.viewportHeight;
    float edgeLen = std::min(num_x, num_y);
    float fovScale = atanf(s.cam.fov / 2);                       // (sic) atan :389
    CamPre c; c.pixelSize = (float)(1.0 / edgeLen * fovScale * 2);
    float w = num_x * c.pixelSize, h = num_y * c.pixelSize;
    c.origin_x = -w / 2; c.origin_y = -h / 2;
    return c;
}
v3 GenRay(Ctx& k, const orc_settings& s, const CamPre& cp, int i, int j) {   // :367-423
    uint32_t local_seed = s.viewportWidth * (uint32_t)j + (uint32_t)i;       // settings.seed ignored (:374)
    Lcg rng; rng.state = tea16(local_seed, s.frameID);
    k.s0 = rng.state; k.pixel_rays = 0;
    float offx = rng.NextFloat() * cp.pixelSize;
    float offy = rng.NextFloat() * cp.pixelSize;
    float u = (i * cp.pixelSize + cp.origin_x + offx);
    float v = (j * cp.pixelSize + cp.origin_y + offy);
    v3 dir = ld3(s.cam.front) + ld3(s.cam.right) * u + ld3(s.cam.up) * v;
    dir = normalize3(dir);
    return LaunchRayRecur(k, ld3(s.cam.pos), dir, (int)s.maxDepth, &rng, 1u);
}

template <class F> void parallel_rows(uint32_t y0, uint32_t y1, int nthreads, F f) {
    if (nthreads <= 1) { for (uint32_t y = y0; y < y1; y++) f(y, 0); return; }
    std::atomic<uint32_t> next(y0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([&, t]() { for (;;) { uint32_t y = next.fetch_add(1); if (y >= y1) break; f(y, t); } });
    for (auto& x : th) x.join();
}

} // namespace

// ------------------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

uint32_t orc_tea16(uint32_t v0, uint32_t v1) { return tea16(v0, v1); }
void orc_lcg_floats(uint32_t state, uint32_t n, float* out) { Lcg l; l.state = state; for (uint32_t i = 0; i < n; i++) out[i] = l.NextFloat(); }
uint32_t orc_node_seed(uint32_t s0, uint32_t node) { return node_seed(s0, node); }

void orc_math_eval(int fn, int math_mode, const float* a, const float* b, uint32_t n, float* out) {
    const MathFns& M = math_mode == ORC_MATH_DET ? kDet : kLibm;
    for (uint32_t i = 0; i < n; i++) {
        float s, c;
        switch (fn) {
            case 0: M.sincos(a[i], &s, &c); out[i] = s; break;
            case 1: M.sincos(a[i], &s, &c); out[i] = c; break;
            case 2: out[i] = M.atan2_(a[i], b[i]); break;
            case 3: out[i] = M.acos_(a[i]); break;
            case 4: out[i] = M.pow_gamma(a[i]); break;
            case 5: out[i] = M.pow5(a[i]); break;
            case 6: out[i] = M.pow10(a[i]); break;
            default: out[i] = 0;
        }
    }
}

static uint32_t copy_mesh(const MeshData& m, orc_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt) {
    if (v) memcpy(v, m.v.data(), sizeof(orc_vertex) * std::min<size_t>(vcap, m.v.size()));
    if (idx) memcpy(idx, m.idx.data(), 4 * std::min<size_t>(icap, m.idx.size()));
    if (icnt) *icnt = (uint32_t)m.idx.size();
    return (uint32_t)m.v.size();
}
uint32_t orc_make_mesh(int kind, const float* xform, orc_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt) {
    MeshData m = kind == 0 ? make_quad() : kind == 1 ? make_cube() : make_sphere();
    M4 T = m4_identity(); if (xform) memcpy(T.d, xform, sizeof T.d);
    m = construct_mesh(m, zero_mat(), T);
    return copy_mesh(m, v, vcap, idx, icap, icnt);
}
uint32_t orc_builtin_scene_mesh_count(int which) { return (uint32_t)builtin(which).size(); }
uint32_t orc_builtin_mesh(int which, uint32_t mesh, orc_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt, orc_material* mat) {
    const MeshData& m = builtin(which)[mesh];
    if (mat) *mat = m.mat;
    return copy_mesh(m, v, vcap, idx, icap, icnt);
}
void orc_builtin_sky(int which, orc_skylight* sky) {             // CornellBox/tracer.cpp:123-128, DefaultScene/tracer.cpp:106-111
    v3 d = -hnormalize3(mk(1, 1, 1));
    sky->color[0] = 0.98f; sky->color[1] = 0.97f; sky->color[2] = 0.99f;
    sky->dir[0] = d.x; sky->dir[1] = d.y; sky->dir[2] = d.z;
    sky->intensity = which == 0 ? 0.f : 5.f;
}
void orc_make_camera(const float pos[3], float azimuth, float zenith, float fov, orc_camera* out) {   // RendererWindow.hpp:15-37
    zenith = zenith < 0.f ? 0.f : (zenith > (float)M_PI ? (float)M_PI : zenith);
    float fy = cosf(zenith), d = sinf(zenith), fx_ = cosf(azimuth), fz_ = sinf(azimuth);
    v3 front = mk(fx_ * d, fy, fz_ * d), right = mk(-fz_, 0, fx_), up = hcross3(right, front);
    memcpy(out->pos, pos, 12); out->fov = fov;
    out->front[0] = front.x; out->front[1] = front.y; out->front[2] = front.z;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
}

orc_scene* orc_scene_create(const orc_scene_desc* d) {
    orc_scene* s = new orc_scene();
    s->sky = d->sky;
    if (d->envMap && d->envW && d->envH) { s->envW = d->envW; s->envH = d->envH; s->env.assign(d->envMap, d->envMap + (size_t)d->envW * d->envH * 3); }
    build_env_cdf(*s);
    uint32_t base = 0;
    for (uint32_t i = 0; i < d->meshCnt; i++) {
        const orc_mesh_desc& md = d->meshes[i];
        OMesh m; m.v.assign(md.vertices, md.vertices + md.vertexCnt); m.idx.assign(md.indices, md.indices + md.indicesCnt);
        m.mat = md.material; m.triBase = base;
        build_mesh_bvh(m);
        uint32_t nt = md.indicesCnt / 3;
        for (uint32_t t = 0; t < nt; t++) { s->triMesh.push_back(i); s->triLocal.push_back(t); }
        base += nt;
        s->meshes.push_back(std::move(m));
    }
    build_emitters(*s);
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
uint32_t orc_scene_triangle_count(const orc_scene* s) { return (uint32_t)s->triMesh.size(); }

int orc_scene_set_flat_bvh(orc_scene* s, const float* nodes, uint32_t n_nodes, const float* tris, uint32_t n_refs) {
    s->fnodes.assign(nodes, nodes + (size_t)n_nodes * 16);
    s->ftris.assign(tris, tris + (size_t)n_refs * 12);
    s->fwidth = 2;
    s->emitRoot = 0; s->nRefsMain = 0;      // (a new tree: orc_scene_set_emitter_tree is called after this, if at all)
    return 0;
}
int orc_scene_set_emitter_tree(orc_scene* s, uint32_t root_ref, uint32_t n_refs_main) {
    const size_t nNodes = s->fnodes.size() / 16, nRefs = s->ftris.size() / 12;
    if (root_ref != 0u) {        // the tree lies behind the main tree in the arrays orc_scene_set_flat_bvh* was given
        if (n_refs_main > nRefs) return 1;
        if (root_ref & 0x80000000u) { const uint32_t cnt = (root_ref >> 27) & 0xFu, first = root_ref & 0x07FFFFFFu; if (first < n_refs_main || (size_t)first + cnt > nRefs) return 1; }
        else if (root_ref >= nNodes) return 1;
    }
    s->emitRoot = root_ref; s->nRefsMain = n_refs_main;
    return 0;
}
int orc_scene_set_estimator(orc_scene* s, int e) { if (e < 0 || e > 3) return 1; s->estimator = e; return 0; }
int orc_scene_set_anyhit_last_level(orc_scene* s, int on) {
    s->anyhitLast = on != 0;
    s->anyEmits = false;
    for (const OMesh& m : s->meshes)
        if (m.mat.emissiveStrength != 0.f && (m.mat.emissiveColor[0] != 0.f || m.mat.emissiveColor[1] != 0.f || m.mat.emissiveColor[2] != 0.f)) s->anyEmits = true;
    return 0;
}
uint64_t orc_scene_get_env_cdf(const orc_scene* s, float* out, uint64_t cap) {
    if (out && cap >= s->envCdf.size()) memcpy(out, s->envCdf.data(), s->envCdf.size() * sizeof(float));
    return s->envCdf.size();
}
uint64_t orc_scene_get_emitters(const orc_scene* s, float* tris, float* cdf, uint64_t cap_tris) {
    const uint64_t n = s->emitTri.size() / 15;
    if (tris && cdf && cap_tris >= n && n) { memcpy(tris, s->emitTri.data(), n * 15 * sizeof(float)); memcpy(cdf, s->emitCdf.data(), (n + 1) * sizeof(float)); }
    return n;
}
int orc_scene_set_flat_width(orc_scene* s, uint32_t width) {
    if (width != 2 && width != 4) return 1;
    s->fwidth = width;
    return 0;
}

// Structural validity of the product's BVH: every reference names a real triangle and carries its exact (v0,e1,e2);
// every triangle is referenced; every leaf box along the path... is checked functionally by the brute-force parity test.
int orc_check_flat_bvh(const orc_scene* s, char* msg, uint32_t cap) {
    auto fail = [&](int code, const std::string& m) { if (msg && cap) snprintf(msg, cap, "%s", m.c_str()); return code; };
    size_t nn = s->fnodes.size() / 16, nr = s->ftris.size() / 12, nt = s->triMesh.size();
    if (nn == 0) return fail(1, "no nodes");
    std::vector<uint8_t> seenTri(nt, 0), seenRef(nr, 0), seenNode(nn, 0);
    std::vector<uint32_t> st; st.push_back(0);
    size_t depthGuard = 0;
    while (!st.empty()) {
        uint32_t cur = st.back(); st.pop_back();
        if (++depthGuard > 4 * (nn + nr) + 16) return fail(2, "cycle in BVH");
        if (cur & 0x80000000u) {
            uint32_t cnt = (cur >> 27) & 0xF, first = cur & 0x07FFFFFFu;
            if ((size_t)first + cnt > nr) return fail(3, "leaf range out of bounds");
            for (uint32_t i = first; i < first + cnt; i++) {
                const float* T = &s->ftris[(size_t)i * 12];
                int32_t g = fbits(T[3]);
                if (g < 0 || (size_t)g >= nt) return fail(4, "bad triangle id in ref");
                const OMesh& m = s->meshes[s->triMesh[g]];
                v3 v0, e1, e2; tri_of(m, s->triLocal[g], &v0, &e1, &e2);
                if (memcmp(&v0, T, 12) || memcmp(&e1, T + 4, 12) || memcmp(&e2, T + 8, 12)) return fail(5, "ref geometry differs from triangle " + std::to_string(g));
                if (fbits(T[7]) != (int32_t)s->triMesh[g]) return fail(6, "ref mesh id mismatch");
                seenTri[g] = 1; seenRef[i] = 1;
            }
        } else {
            if (cur >= nn) return fail(7, "node index out of bounds");
            if (seenNode[cur]) return fail(8, "node visited twice");
            seenNode[cur] = 1;
            const float* N = &s->fnodes[(size_t)cur * 16];
            st.push_back((uint32_t)fbits(N[12])); st.push_back((uint32_t)fbits(N[13]));
            if (s->fwidth == 4) {
                st.push_back((uint32_t)fbits(N[14])); st.push_back((uint32_t)fbits(N[15]));
                const uint32_t sb[3] = { (uint32_t)fbits(N[10]), (uint32_t)fbits(N[11]), (uint32_t)fbits(N[3]) };      // scale x, y, z
                for (int a = 0; a < 3; a++) { uint32_t e = sb[a] >> 23; if ((sb[a] & 0x807FFFFFu) || e < 67 || e > 167) return fail(11, "quantisation scale out of range"); }
                for (int i = 0; i < 4; i++) for (int a = 0; a < 3; a++) {
                    uint32_t lo = ((uint32_t)fbits(N[4 + a]) >> (8 * i)) & 0xFFu, hi = ((uint32_t)fbits(N[7 + a]) >> (8 * i)) & 0xFFu;
                    const bool empty = (uint32_t)fbits(N[12 + i]) == 0x80000000u;
                    if (!empty && lo > hi) return fail(12, "inverted quantised child box");
                    if (empty && !(lo == 255u && hi == 0u)) return fail(13, "empty child slot without an inverted box");
                }
            }
        }
    }
    for (size_t t = 0; t < nt; t++) if (!seenTri[t]) return fail(9, "triangle " + std::to_string(t) + " not referenced");
    for (size_t n = 0; n < nn; n++) if (!seenNode[n]) return fail(10, "unreachable node");
    return 0;
}

int orc_trace_rays(const orc_scene* s, int mode, const float* origins, const float* dirs, uint32_t n, orc_hit* out, orc_stats* stats, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> cs(nthreads);
    const uint32_t chunk = 1024; uint32_t nchunks = (n + chunk - 1) / chunk;
    parallel_rows(0, nchunks, nthreads, [&](uint32_t ci, int t) {
        uint32_t b = ci * chunk, e = std::min(n, b + chunk);
        for (uint32_t i = b; i < e; i++) {
            Hit h = isect(*s, mode, ld3(origins + (size_t)i * 3), ld3(dirs + (size_t)i * 3), &cs[t]);
            orc_hit& o = out[i];
            o.t = h.t; o.u = h.u; o.v = h.v; o.tri = h.tri; o.mesh = h.tri >= 0 ? (int32_t)s->triMesh[h.tri] : -1; o.front = h.front ? 1 : 0;
        }
    });
    if (stats) { memset(stats, 0, sizeof *stats); for (auto& c : cs) { stats->rays += c.rays; stats->node_visits += c.nodes; stats->tri_tests += c.tris; stats->hits += c.hits; stats->shadow_rays += c.shadow; stats->anyhit_rays += c.anyhit; stats->emitfirst_rays += c.emitfirst; } }
    return 0;
}

int orc_render(const orc_scene* s, const orc_settings* st, int isect_mode, int rng_mode, int math_mode,
               uint32_t first_frame, uint32_t n_frames, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               float* acc, uint8_t* ldr, float* sample, uint32_t* rays_per_pixel, orc_stats* stats, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const uint32_t W = st->viewportWidth, H = st->viewportHeight;
    if (x1 > W) x1 = W; if (y1 > H) y1 = H;
    const MathFns& M = math_mode == ORC_MATH_DET ? kDet : kLibm;
    std::vector<Counters> cs(nthreads);
    for (uint32_t f = first_frame; f < first_frame + n_frames; f++) {
        orc_settings cur = *st; cur.frameID = f;
        CamPre cp = cam_pre(cur);
        float weight = (float)(1.0 / (f + 1.0));                  // CUDATracerProg.cu:878
        // work items = 64-pixel spans of a row, so that even a one-row rectangle keeps every thread busy
        const uint32_t span = 64, spansPerRow = (x1 - x0 + span - 1) / span;
        parallel_rows(0, (y1 - y0) * spansPerRow, nthreads, [&](uint32_t item, int t) {
            const uint32_t j = y0 + item / spansPerRow, xs = x0 + (item % spansPerRow) * span, xe = std::min(x1, xs + span);
            Ctx k; k.s = s; k.isect = isect_mode; k.rng_mode = rng_mode; k.M = &M; k.c = &cs[t]; k.s0 = 0; k.pixel_rays = 0;
            for (uint32_t i = xs; i < xe; i++) {
                v3 c = GenRay(k, cur, cp, (int)i, (int)j);
                size_t addr = ((size_t)j * W + i) * 4;
                if (sample) { size_t a3 = ((size_t)j * W + i) * 3; sample[a3] = c.x; sample[a3 + 1] = c.y; sample[a3 + 2] = c.z; }
                if (rays_per_pixel) rays_per_pixel[(size_t)j * W + i] = k.pixel_rays;
                float r = acc[addr] * (1 - weight) + c.x * weight;     // :895-897
                float g = acc[addr + 1] * (1 - weight) + c.y * weight;
                float b = acc[addr + 2] * (1 - weight) + c.z * weight;
                acc[addr] = r; acc[addr + 1] = g; acc[addr + 2] = b; acc[addr + 3] = 1;
                if (ldr) {                                              // :908-919
                    float cc[3] = { saturatef(r), saturatef(g), saturatef(b) };
                    for (int q = 0; q < 3; q++) { float x = cc[q] / (cc[q] + 1.0f); x = M.pow_gamma(x); ldr[addr + q] = (uint8_t)(int)(x * 255); }
                    ldr[addr + 3] = 255;
                }
            }
        });
    }
    if (stats) { memset(stats, 0, sizeof *stats); for (auto& c : cs) { stats->rays += c.rays; stats->node_visits += c.nodes; stats->tri_tests += c.tris; stats->hits += c.hits; stats->shadow_rays += c.shadow; stats->anyhit_rays += c.anyhit; stats->emitfirst_rays += c.emitfirst; } }
    return 0;
}

} // extern "C"
