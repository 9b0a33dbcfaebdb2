// kernels.hip -- gfx950 kernels of the path-tracing core.
//
// Replaces the reference's single megakernel `GenRay` (CUDATracer/CUDATracerProg/CUDATracerProg.cu:859-922):
//   k_wavestack  the default integrator: persistent waves, each with a private LIFO of pending rays in HBM, in-wave
//                refill of idle lanes, a two-ended LDS shade queue and dense shade batches; one launch per frame.
//                Template flags: scene traversed from LDS, quantised 4-wide nodes, light-sampling estimator.
//   k_level      the first design, one launch per recursion level with global ping-pong queues (integrator = 1)
//   k_dfs        one lane per pixel-sample in the reference's depth-first order (rng_mode = dfs), bit-identical to it
//   k_accumulate / k_tonemap / k_trace_rays / k_test_math
// In the queue-based integrators radiance is carried top-down as a throughput weight, so no return values flow back up
// the recursion tree (the reference's recursion, CUDATracerProg.cu:426-541, is linear in the children's radiance).
//
// Per-function reference lines are cited inline.  Arithmetic contract: see device_math.h.
#include "kernels.h"
#include "device_math.h"

#include <float.h>

namespace mipt {

constexpr int kBlock = 256;
constexpr int kStackStride = kBlock;   // dword stride between a lane's LDS stack entries (k_wavestack shadows it with its own block size)
constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kTerm = 0xFFFFFFFFu;
constexpr int kStackCap = 64;      // == kMaxBvhDepth of the builder
constexpr int kWideStackCap = 96;  // == kMaxWideStack of the builder (4-wide nodes push up to three entries per level)
constexpr int kLdsStack = 12;      // entries per lane kept in LDS; deeper entries spill to scratch

struct HitRec { float t, u, v; int ref; int tri; bool front; };   // ref = reference index, tri = global triangle id

struct RayPre { v3 o, d, idir, ood; };

// CUDATracerProg.cu:567-574
// SHORT_RCP: the reciprocals through rcp_newton<1> (3 instructions instead of the 11 of an IEEE division; bit-identical for
// magnitudes in [2^-100, 2^100], mipt_test_rcp_sweep) -- for directions the integrator generated itself, which are of
// unit scale; caller-supplied rays (k_trace_rays) keep the full division.
template <bool SHORT_RCP = false>
MIPT_DEV RayPre ray_pre(v3 o, v3 d) {
    const float ooeps = 8.271806125530277e-25f;   // exp2f(-80)
    RayPre r; r.o = o; r.d = d;
    const float gx = __builtin_fabsf(d.x) > ooeps ? d.x : __builtin_copysignf(ooeps, d.x);
    const float gy = __builtin_fabsf(d.y) > ooeps ? d.y : __builtin_copysignf(ooeps, d.y);
    const float gz = __builtin_fabsf(d.z) > ooeps ? d.z : __builtin_copysignf(ooeps, d.z);
    if constexpr (SHORT_RCP) { r.idir.x = rcp_newton<1>(gx); r.idir.y = rcp_newton<1>(gy); r.idir.z = rcp_newton<1>(gz); }
    else { r.idir.x = 1.0f / gx; r.idir.y = 1.0f / gy; r.idir.z = 1.0f / gz; }
    r.ood = mk3(o.x * r.idir.x, o.y * r.idir.y, o.z * r.idir.z);
    return r;
}

// slab test: CUDATracerProg.cu:651-658 + spanBegin/EndKepler (:86-87) -> v_min3/v_max3 on gfx950
MIPT_DEV bool slab(const RayPre& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float hitT, float& tnear) {
    float lox = ffma(mnx, r.idir.x, -r.ood.x), hix = ffma(mxx, r.idir.x, -r.ood.x);
    float loy = ffma(mny, r.idir.y, -r.ood.y), hiy = ffma(mxy, r.idir.y, -r.ood.y);
    float loz = ffma(mnz, r.idir.z, -r.ood.z), hiz = ffma(mxz, r.idir.z, -r.ood.z);
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lox, hix), __builtin_fminf(loy, hiy)),
                                 __builtin_fmaxf(__builtin_fminf(loz, hiz), 0.0f));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lox, hix), __builtin_fmaxf(loy, hiy)),
                                 __builtin_fminf(__builtin_fmaxf(loz, hiz), hitT));
    tnear = tmin;
    return tmin <= tmax;
}

// Moller-Trumbore on a precomputed (v0, e1, e2) reference: CUDATracerProg.cu:232-263.  Strict t < tmin as in the
// reference; an exact tie goes to the lower global triangle id (the reference lets the first triangle found win, which
// across meshes is the lower mesh index), so the closest hit does not depend on the BVH's shape.
// SHORT_RCP: 1 / det through rcp_newton<1> (see ray_pre): |det| >= 2^-23 here, and <= 2^84 for unit-scale directions and
// the vertex coordinates mipt_scene_create accepts (below 2^40).
template <bool SHORT_RCP = false>
MIPT_DEV bool mt_test(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float tmin, int id, int bestId, float& t_, float& u_, float& v_, bool& front) {
    v3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (__builtin_fabsf(det) < FLT_EPSILON) return false;
    float invDet = SHORT_RCP ? rcp_newton<1>(det) : 1 / det;
    v3 tvec = o - v0;
    float u = dot3(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    v3 qvec = cross3(tvec, e1);
    float v = dot3(d, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    float t = dot3(e2, qvec) * invDet;
    if (t > 0 && (t < tmin || (t == tmin && id < bestId))) { t_ = t; u_ = u; v_ = v; front = det > 0; return true; }
    return false;
}

// Two triangles per test with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: each half is the IEEE
// operation of the scalar form, so every value below equals what mt_test computes for that triangle, bit for bit).  The
// pair comes interleaved from SceneDev::refs2, so that a loaded float4 holds (a, b) of two components in adjacent
// registers; the ray's components are broadcast by the instruction's operand select.
typedef float v2f __attribute__((ext_vector_type(2)));
MIPT_DEV v2f pk(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
MIPT_DEV v2f pk1(float a) { v2f r; r.x = a; r.y = a; return r; }
MIPT_DEV v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
struct v3p { v2f x, y, z; };
MIPT_DEV v2f dot3p(const v3p& a, const v3p& b) { return pkfma(a.z, b.z, pkfma(a.y, b.y, a.x * b.x)); }
MIPT_DEV v3p cross3p(const v3p& a, const v3p& b) {
    v3p r;
    r.x = pkfma(a.y, b.z, -(b.y * a.z)); r.y = pkfma(a.z, b.x, -(b.z * a.x)); r.z = pkfma(a.x, b.y, -(b.x * a.y));
    return r;
}
struct PairTest { v2f det, t, u, v, uv; };
#ifdef MIPT_EXP_PAIR_SCALAR
// diagnostic: the same pair layout and loop, the arithmetic of the two halves as separate scalar instructions
MIPT_DEV void mt_half(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float& det, float& t, float& u, float& v, float& uv) {
    v3 pvec = cross3(d, e2);
    det = dot3(e1, pvec);
    const float inv = rcp_newton<1>(det);
    v3 tvec = o - v0;
    u = dot3(tvec, pvec) * inv;
    v3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * inv;
    t = dot3(e2, qvec) * inv;
    uv = u + v;
}
MIPT_DEV PairTest mt_pair(v3 o, v3 d, const v3p& v0, const v3p& e1, const v3p& e2) {
    PairTest r; float det, t, u, v, uv;
    mt_half(o, d, mk3(v0.x.x, v0.y.x, v0.z.x), mk3(e1.x.x, e1.y.x, e1.z.x), mk3(e2.x.x, e2.y.x, e2.z.x), det, t, u, v, uv);
    r.det.x = det; r.t.x = t; r.u.x = u; r.v.x = v; r.uv.x = uv;
    mt_half(o, d, mk3(v0.x.y, v0.y.y, v0.z.y), mk3(e1.x.y, e1.y.y, e1.z.y), mk3(e2.x.y, e2.y.y, e2.z.y), det, t, u, v, uv);
    r.det.y = det; r.t.y = t; r.u.y = u; r.v.y = v; r.uv.y = uv;
    return r;
}
#else
MIPT_DEV PairTest mt_pair(v3 o, v3 d, const v3p& v0, const v3p& e1, const v3p& e2) {
    v3p D; D.x = pk1(d.x); D.y = pk1(d.y); D.z = pk1(d.z);
    v3p pvec = cross3p(D, e2);
    PairTest r;
    r.det = dot3p(e1, pvec);
    v2f inv = pk(__builtin_amdgcn_rcpf(r.det.x), __builtin_amdgcn_rcpf(r.det.y));       // rcp_newton<1>, the correction packed
    const v2f e = pkfma(-r.det, inv, pk1(1.0f));
    inv = pkfma(e, inv, inv);
    v3p tvec; tvec.x = pk1(o.x) - v0.x; tvec.y = pk1(o.y) - v0.y; tvec.z = pk1(o.z) - v0.z;
    r.u = dot3p(tvec, pvec) * inv;
    v3p qvec = cross3p(tvec, e1);
    r.v = dot3p(D, qvec) * inv;
    r.t = dot3p(e2, qvec) * inv;
    r.uv = r.u + r.v;
    return r;
}
#endif
// the accept / reject decisions of mt_test on one half of a PairTest (same comparisons, same NaN behaviour)
// (one flat expression: in a wave some lane nearly always passes each early test, so branches only add mask bookkeeping)
MIPT_DEV bool mt_accept(float det, float t, float u, float v, float uv, float tmin, int id, int bestId) {
    const bool reject = (__builtin_fabsf(det) < FLT_EPSILON) | (u < 0) | (u > 1) | (v < 0) | (uv > 1);
    const bool closer = (t > 0) & ((t < tmin) | ((t == tmin) & (id < bestId)));
    return closer & !reject;
}
// does the ray hit the triangle of a reference record at all (t > 0)?  The leaf loop's arithmetic and acceptance, with no bound
MIPT_DEV bool mt_hits(v3 o, v3 d, float4 a, float4 b, float4 c) {
    const v3 v0_ = xyz(a), e1_ = xyz(b), e2_ = xyz(c);
    const v3 pvec = cross3(d, e2_);
    const float det = dot3(e1_, pvec);
    const float inv = rcp_newton<1>(det);
    const v3 tvec = o - v0_;
    const float uu = dot3(tvec, pvec) * inv;
    const v3 qvec = cross3(tvec, e1_);
    const float vv = dot3(d, qvec) * inv;
    const float tt = dot3(e2_, qvec) * inv;
    return mt_accept(det, tt, uu, vv, uu + vv, __builtin_inff(), 0, -1);
}

// Traversal stack: the first kLdsStack entries of every lane live in LDS (bank = lane, conflict-free), deeper
// entries in a private scratch array that is only touched by unusually deep paths.  The stack pointer is a plain
// register and the two halves are reached through separate branches: folding them into one struct/pointer select
// puts the pointer itself in scratch and turns every pop into a flat load behind a scratch load.
#define MIPT_STACK_PUSH(lds_, deep_, sp_, x_)                                              \
    do { if ((sp_) < kLdsStack) (lds_)[(sp_) * kStackStride] = (x_); else (deep_)[(sp_) - kLdsStack] = (x_); (sp_)++; } while (0)
#define MIPT_STACK_POP(lds_, deep_, sp_, out_)                                             \
    do { (sp_)--;                                                                            \
         if ((sp_) < kLdsStack) { (out_) = (lds_)[(sp_) * kStackStride]; asm volatile("" : "+v"(out_)); }      /* keep ds_read ... */ \
         else { (out_) = (deep_)[(sp_) - kLdsStack]; asm volatile("" : "+v"(out_)); }                    /* ... and scratch_load apart */ \
    } while (0)

// LDS-resident scenes are only traversed this way when the builder's stack bound fits the LDS part (the host checks
// SceneDev::stackNeed <= kLdsStack), so their pushes and pops need no scratch branch at all.
#define MIPT_STACK_PUSH_L(lds_, sp_, x_) do { (lds_)[(sp_) * kStackStride] = (x_); (sp_)++; } while (0)
#define MIPT_STACK_POP_L(lds_, sp_, out_) do { (sp_)--; (out_) = (lds_)[(sp_) * kStackStride]; } while (0)

// Closest hit (TraceRay, CUDATracerProg.cu:543-794) on the flat scene BVH.  Near child first, far child pushed;
// leaves are tested when they become current; hitT shrinks as hits are found and culls later box tests.
// With LDS_SCENE the whole node + reference stream has been copied into LDS by the block (scenes of a few KB, e.g. the
// Cornell box): `nodes`/`refs` then point into LDS and every dependent fetch of the walk costs an LDS round trip
// instead of an L1/L2 one.
template <typename NodePtr, typename RefPtr>
MIPT_DEV HitRec traverse(NodePtr nodes, RefPtr refs, v3 o, v3 d, uint32_t* ldsStack) {
    RayPre r = ray_pre(o, d);
    HitRec h; h.t = __builtin_inff(); h.u = 0.f; h.v = 0.f; h.ref = -1; h.tri = -1; h.front = false;
    uint32_t deep[kStackCap - kLdsStack];
    int tsp = 0;
    MIPT_STACK_PUSH(ldsStack, deep, tsp, kTerm);
    uint32_t cur = 0;
    while (true) {
        while (!(cur & kLeaf)) {
            auto N = nodes + (size_t)cur * 4;
            float4 n0 = N[0], n1 = N[1], n2 = N[2], n3 = N[3];
            float t0, t1;
            bool h0 = slab(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, h.t, t0);
            bool h1 = slab(r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, h.t, t1);
            uint32_t c0 = __float_as_uint(n3.x), c1 = __float_as_uint(n3.y);
            if (h0 || h1) {
                uint32_t first = h0 ? c0 : c1;
                if (h0 && h1) {
                    uint32_t second = c1;
                    if (t1 < t0) { first = c1; second = c0; }
                    MIPT_STACK_PUSH(ldsStack, deep, tsp, second);
                }
                cur = first;
            } else {
                MIPT_STACK_POP(ldsStack, deep, tsp, cur);
            }
        }
        if (cur == kTerm) break;
        uint32_t cnt = (cur >> 27) & 0xFu, first = cur & 0x07FFFFFFu;
        for (uint32_t i = first; i < first + cnt; i++) {
            auto T = refs + (size_t)i * 3;
            float4 a = T[0], b = T[1], c = T[2];
            float tt, uu, vv; bool ff;
            int id = (int)__float_as_uint(a.w);
            if (mt_test(o, d, xyz(a), xyz(b), xyz(c), h.t, id, h.tri, tt, uu, vv, ff)) { h.t = tt; h.u = uu; h.v = vv; h.front = ff; h.ref = (int)i; h.tri = id; }
        }
        MIPT_STACK_POP(ldsStack, deep, tsp, cur);
    }
    return h;
}

// One visit of a quantised 4-wide node (layout: bvh_builder.h).  The ray is moved into the node's quantised frame
// (t = q * (idir * scale) + (origin - o) * idir), so a child plane costs one byte->float conversion and one FMA.
// Children that are hit come back sorted by entry distance (s0 nearest); returns how many were hit.
#define MIPT_CE(ta_, ra_, tb_, rb_) do { if ((tb_) < (ta_)) { float t_ = (ta_); (ta_) = (tb_); (tb_) = t_; uint32_t r_ = (ra_); (ra_) = (rb_); (rb_) = r_; } } while (0)
MIPT_DEV int wide_visit(const RayPre& r, float4 n0, float4 n1, float4 n2, float4 n3, float hitT,
                        uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3) {
    const float ix = r.idir.x * n2.z, iy = r.idir.y * n2.w, iz = r.idir.z * n0.w;      // the node's scales are floats
    // (the FMA form n0 * idir - o * idir saves three instructions and measured no gain: profiles/r2/exp_visit_isa.log)
    const float ox = (n0.x - r.o.x) * r.idir.x, oy = (n0.y - r.o.y) * r.idir.y, oz = (n0.z - r.o.z) * r.idir.z;
    // (picking entry / exit planes by the ray's octant -- one select per axis for all four children instead of a
    // min/max per plane pair -- gives the same values and was measured 5 % slower: the compiler already folds these
    // into v_min3/v_max3)
    const uint32_t lx = __float_as_uint(n1.x), ly = __float_as_uint(n1.y), lz = __float_as_uint(n1.z);
    const uint32_t hx = __float_as_uint(n1.w), hy = __float_as_uint(n2.x), hz = __float_as_uint(n2.y);
    float t[4]; uint32_t ref[4] = { __float_as_uint(n3.x), __float_as_uint(n3.y), __float_as_uint(n3.z), __float_as_uint(n3.w) };
    int n = 0;
    // Entry / exit planes picked by the ray's octant: one select per axis word serves all four children, and a child
    // costs max3 / min3 instead of three min + three max.  Same values as min(lo, hi) / max(lo, hi): q_lo <= q_hi and the
    // FMA is monotonic in q, so for idir > 0 the low plane is the entry plane and for idir < 0 the high one.
    const bool ngx = r.idir.x < 0.f, ngy = r.idir.y < 0.f, ngz = r.idir.z < 0.f;
    const uint32_t ex = ngx ? hx : lx, fx = ngx ? lx : hx;
    const uint32_t ey = ngy ? hy : ly, fy = ngy ? ly : hy;
    const uint32_t ez = ngz ? hz : lz, fz = ngz ? lz : hz;
    // (one v_pk_fma_f32 per axis for the entry + exit plane of a child -- 12 packed FMAs instead of 24 plain ones, clean
    // code, no extra moves -- measured 1.5 % SLOWER on Default and the grid: profiles/r2/exp_rejected_variants.log)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sh = 8 * i;
        const float tex = ffma((float)((ex >> sh) & 0xFFu), ix, ox), tfx = ffma((float)((fx >> sh) & 0xFFu), ix, ox);
        const float tey = ffma((float)((ey >> sh) & 0xFFu), iy, oy), tfy = ffma((float)((fy >> sh) & 0xFFu), iy, oy);
        const float tez = ffma((float)((ez >> sh) & 0xFFu), iz, oz), tfz = ffma((float)((fz >> sh) & 0xFFu), iz, oz);
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(tex, tey), __builtin_fmaxf(tez, 0.0f));
        const float tmax = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, hitT));
        const bool hit = tmin <= tmax;        // an empty slot holds an inverted box: no test of its reference needed
        t[i] = hit ? tmin : __builtin_inff();
        n += hit ? 1 : 0;
    }
    MIPT_CE(t[0], ref[0], t[1], ref[1]); MIPT_CE(t[2], ref[2], t[3], ref[3]);
    MIPT_CE(t[0], ref[0], t[2], ref[2]); MIPT_CE(t[1], ref[1], t[3], ref[3]);
    MIPT_CE(t[1], ref[1], t[2], ref[2]);
    s0 = ref[0]; s1 = ref[1]; s2 = ref[2]; s3 = ref[3];
    return n;
}

// Closest hit on the 4-wide layout: nearest hit child first, the others pushed farthest-first.
template <typename NodePtr, typename RefPtr>
MIPT_DEV HitRec traverse_wide(NodePtr nodes, RefPtr refs, v3 o, v3 d, uint32_t* ldsStack) {
    RayPre r = ray_pre(o, d);
    HitRec h; h.t = __builtin_inff(); h.u = 0.f; h.v = 0.f; h.ref = -1; h.tri = -1; h.front = false;
    uint32_t deep[kWideStackCap - kLdsStack];
    int tsp = 0;
    MIPT_STACK_PUSH(ldsStack, deep, tsp, kTerm);
    uint32_t cur = 0;
    while (true) {
        while (!(cur & kLeaf)) {
            auto N = nodes + (size_t)cur * 4;
            float4 n0 = N[0], n1 = N[1], n2 = N[2], n3 = N[3];
            uint32_t s0, s1, s2, s3;
            const int n = wide_visit(r, n0, n1, n2, n3, h.t, s0, s1, s2, s3);
            if (n == 0) { MIPT_STACK_POP(ldsStack, deep, tsp, cur); }
            else {
                if (n > 3) MIPT_STACK_PUSH(ldsStack, deep, tsp, s3);
                if (n > 2) MIPT_STACK_PUSH(ldsStack, deep, tsp, s2);
                if (n > 1) MIPT_STACK_PUSH(ldsStack, deep, tsp, s1);
                cur = s0;
            }
        }
        if (cur == kTerm) break;
        uint32_t cnt = (cur >> 27) & 0xFu, first = cur & 0x07FFFFFFu;
        for (uint32_t i = first; i < first + cnt; i++) {
            auto T = refs + (size_t)i * 3;
            float4 a = T[0], b = T[1], c = T[2];
            float tt, uu, vv; bool ff;
            int id = (int)__float_as_uint(a.w);
            if (mt_test(o, d, xyz(a), xyz(b), xyz(c), h.t, id, h.tri, tt, uu, vv, ff)) { h.t = tt; h.u = uu; h.v = vv; h.front = ff; h.ref = (int)i; h.tri = id; }
        }
        MIPT_STACK_POP(ldsStack, deep, tsp, cur);
    }
    return h;
}

// ---- sampling (CUDATracerProg.cu:93-180) --------------------------------------------------------------------
MIPT_DEV void cone_basis(v3 dir, v3& right, v3& up) {
    float rx = dir.z, ry = -dir.x;
    float rlen = __builtin_sqrtf(0.f + rx * rx + ry * ry);
    right = rlen <= 0 ? mk3(0, 0, 1) : mk3(rx / rlen, 0.f, ry / rlen);      // (0 / rlen is +0 for rlen > 0: no division spent on it)
    up = cross3(right, dir);
}
// (the basis comes from the caller: one bounce draws twice about the same normal, and the compiler did not share the two
//  inlined copies of cone_basis across the opacity branch -- a square root and two divisions per shaded hit)
MIPT_DEV v3 rand_dir_inside_cone(Lcg& rng, v3 dir, v3 right, v3 up, float size) {
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(6.283185307179586 * (double)u);
    float sx = 1 - 2 * v * size;
    float sn, cs; det_sincosf(theta, &sn, &cs);
    float sy = sqrt_unit(1 - sx * sx) * cs;
    float sz = sqrt_unit(1 - sx * sx) * sn;
    return sx * dir + sy * up + sz * right;
}
MIPT_DEV v3 rand_dir_inside_cone(Lcg& rng, v3 dir, float size) {
    v3 right, up; cone_basis(dir, right, up);
    return rand_dir_inside_cone(rng, dir, right, up, size);
}
MIPT_DEV v3 ggx_rand_dir_inside_cone(Lcg& rng, v3 dir, v3 right, v3 up, float size) {
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(6.283185307179586 * (double)u);
    float phi = det_atan2f(size * sqrt_unit(v), sqrt_unit(1 - v));
    float sp, cp; det_sincosf(phi, &sp, &cp);
    float sx = 1 - sp;                                        // (sic) CUDATracerProg.cu:160
    float sn, cs; det_sincosf(theta, &sn, &cs);
    float sy = sqrt_unit(1 - sx * sx) * cs;
    float sz = sqrt_unit(1 - sx * sx) * sn;
    return sx * dir + sy * up + sz * right;
}
MIPT_DEV v3 ggx_rand_dir_inside_cone(Lcg& rng, v3 dir, float size) {
    v3 right, up; cone_basis(dir, right, up);
    return ggx_rand_dir_inside_cone(rng, dir, right, up, size);
}

// ---- estimator 2 (include/mipt.h): the same two lobes sampled the textbook way ------------------------------------
// diffuse: cosine-weighted hemisphere about N, same basis and draws as rand_dir_inside_cone
MIPT_DEV v3 cosine_dir_about(Lcg& rng, v3 N) {
    v3 right, up; cone_basis(N, right, up);
    float u = rng.NextFloat(), v = rng.NextFloat();
    float theta = (float)(6.283185307179586 * (double)u);
    float sn, cs; det_sincosf(theta, &sn, &cs);
    float r = __builtin_sqrtf(v), cz = __builtin_sqrtf(1 - v);
    return cz * N + (r * cs) * up + (r * sn) * right;
}
// specular: microfacet normal from the GGX distribution of visible normals (Heitz 2018), alpha = roughness
MIPT_DEV v3 ggx_visible_normal(Lcg& rng, v3 N, v3 d, float alpha) {
    float u1 = rng.NextFloat(), u2 = rng.NextFloat();
    if (!(alpha > 0.f)) return N;
    v3 right, up; cone_basis(N, right, up);
    v3 V = -d;
    v3 Ve = mk3(dot3(V, up), dot3(V, right), __builtin_fmaxf(dot3(V, N), 1e-4f));
    v3 Vh = normalize3(mk3(alpha * Ve.x, alpha * Ve.y, Ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.f ? mk3(-Vh.y, Vh.x, 0.f) / __builtin_sqrtf(lensq) : mk3(1.f, 0.f, 0.f);
    v3 T2 = cross3(Vh, T1);
    float r = __builtin_sqrtf(u1);
    float phi = (float)(6.283185307179586 * (double)u2);
    float sn, cs; det_sincosf(phi, &sn, &cs);
    float t1 = r * cs, t2 = r * sn;
    float sw = 0.5f * (1 + Vh.z);
    t2 = (1 - sw) * __builtin_sqrtf(1 - t1 * t1) + sw * t2;
    float nz = __builtin_sqrtf(__builtin_fmaxf(0.f, 1 - t1 * t1 - t2 * t2));
    v3 Nh = T1 * t1 + T2 * t2 + Vh * nz;
    v3 Ne = normalize3(mk3(alpha * Nh.x, alpha * Nh.y, __builtin_fmaxf(1e-6f, Nh.z)));
    return Ne.x * up + Ne.y * right + Ne.z * N;
}
MIPT_DEV float ggx_g1(float nl, float alpha) {
    if (!(nl > 0.f)) return 0.f;
    float a2 = alpha * alpha;
    return (2 * nl) / (nl + __builtin_sqrtf(a2 + (1 - a2) * (nl * nl)));
}

// ---- miss (OnMiss, CUDATracerProg.cu:796-846) ------------------------------------------------------------------
MIPT_DEV v3 on_miss(const SceneDev& S, v3 d) {
    if (S.env == nullptr) {
        v3 sd = mk3(S.skyDir[0], S.skyDir[1], S.skyDir[2]);
        float ang = (1 - dot3(d, sd)) / 2;
        return mk3(S.skyColor[0], S.skyColor[1], S.skyColor[2]) * S.skyIntensity * pow10_mul(ang);
    }
    float dy = __builtin_fminf(1.0f, __builtin_fmaxf(-1.0f, d.y));
    float zenith = det_acosf(dy);
    float azimuth = det_atan2f(d.x, d.z);
    float v = (float)((double)zenith / 3.14159265358979323846);
    float u = (float)((double)azimuth / (3.14159265358979323846 * 2) + 0.5);
    u *= (float)S.envW; v *= (float)S.envH;
    int u0 = (int)u, v0 = (int)v;
    int uc = min(max(u0, 0), (int)S.envW - 1), vc = min(max(v0, 0), (int)S.envH - 1);
    uint32_t a = (uint32_t)vc * S.envW + (uint32_t)uc;
    const float* px = S.env + (size_t)a * 3;
    v3 p00 = mk3(px[0], px[1], px[2]);
    float du = u - u0, dv = v - v0;
    v3 p0 = (dv) * p00 + (1 - dv) * p00;
    v3 p1 = (dv) * p00 + (1 - dv) * p00;
    return du * p1 + (1 - du) * p0;
}

struct Material { v3 color; float roughness, metallic, opacity, n, emissiveStrength; v3 emissive; float r0; };
MIPT_DEV Material load_material(const SceneDev& S, uint32_t mesh) {
    const float4* M = S.mats + (size_t)mesh * 3;
    float4 a = M[0], b = M[1], c = M[2];
    Material m; m.color = xyz(a); m.roughness = a.w; m.metallic = b.x; m.opacity = b.y; m.n = b.z; m.emissiveStrength = b.w; m.emissive = xyz(c); m.r0 = c.w;
    return m;
}

// Surface frame at the hit (tail of NearestIntersectSubset, CUDATracerProg.cu:268-294)
struct Surface { v3 P, N; uint32_t mesh; };
MIPT_DEV Surface surface_at(const SceneDev& S, v3 o, v3 d, const HitRec& h) {
    const float4* Sh = S.shade + (size_t)h.tri * 3;
    float4 s0 = Sh[0], s1 = Sh[1], s2 = Sh[2];
    float w0 = 1 - h.u - h.v, w1 = h.u, w2 = h.v;
    v3 hn = xyz(s0) * w0 + xyz(s1) * w1 + xyz(s2) * w2;
    Surface sf; sf.P = o + d * h.t; sf.N = h.front ? hn : -hn; sf.mesh = __float_as_uint(s0.w);
    return sf;
}

// One bounce of LaunchRayRecur (CUDATracerProg.cu:439-528), top-down: given the incoming ray and the hit, produce the
// two child rays and the factors their radiance is multiplied by.  `rng` supplies GGX (2) then diffuse (2) draws.
struct Bounce { v3 o1, d1, f1; v3 o2, d2, f2; v3 emit; };
template <bool PBS = false>      // PBS: estimator 2
MIPT_DEV Bounce shade_hit(const Material& mat, const Surface& sf, v3 d, bool front, Lcg& rng) {
    Bounce b;
    v3 N = sf.N, P = sf.P;
    v3 right, up;
    if constexpr (!PBS) cone_basis(N, right, up);
    v3 mN = PBS ? ggx_visible_normal(rng, N, d, mat.roughness) : ggx_rand_dir_inside_cone(rng, N, right, up, mat.roughness);
    v3 R = reflect3(d, mN);
    b.o1 = P + N * 1e-6f; b.d1 = R;
    float n1 = 1.f, n2 = mat.n; n2 += FLT_EPSILON;
    if (!front) { float t = n1; n1 = n2; n2 = t; }
    const float r0 = mat.r0;      // ((n1 - n2) / (n1 + n2))^2: the same for either order of n1, n2, so computed once per material on the host (api.hip fill_materials)
    float rr = (1 - dot3(-d, mN));
    rr = pow5_mul(rr);
    rr = r0 + (1 - r0) * rr;
    rr = saturatef(rr);
    rr = mat.metallic * 1.f + (1 - mat.metallic) * rr;
    float rt = 1 - rr;
    v3 absorb = mat.metallic * mat.color + (1 - mat.metallic) * mk3(1.f, 1.f, 1.f);
    b.f1 = absorb * rr;
    if constexpr (PBS) b.f1 = b.f1 * ggx_g1(dot3(R, N), mat.roughness);
    if (mat.opacity < 1.f) {
        v3 frontv = (d + R) / 2.0f;
        v3 down = d - frontv;
        float down_len = length3(down);
        down = down_len < 1e-10f ? mk3(1, 0, 0) : (down / down_len);
        v3 front_refr = frontv * (n1 / n2);
        float frl = length3(front_refr);
        if (frl > 1) frl = 1;
        float drl = __builtin_sqrtf(1 - frl * frl);
        b.d2 = normalize3(front_refr + down * drl);
        b.o2 = P - N * 1e-6f;
    } else {
        b.d2 = PBS ? cosine_dir_about(rng, N) : rand_dir_inside_cone(rng, N, right, up, 0.49f);
        b.o2 = P + N * 1e-6f;
    }
    b.f2 = mat.color * rt * ((mat.opacity == 1.f) ? 1.f : (1 - mat.opacity));
    b.emit = mat.emissive * mat.emissiveStrength;
    return b;
}

// Primary ray (RTProgram::GenRay, CUDATracerProg.cu:367-417)
MIPT_DEV void gen_primary(const FrameDev& F, uint32_t pix, v3& o, v3& d, uint32_t& s0) {
    uint32_t i = pix % F.W, j = pix / F.W;
    Lcg rng; rng.state = tea16(pix, F.frameID);      // local_seed = w*j + i (:373); settings.seed unused (:374)
    s0 = rng.state;
    float offx = rng.NextFloat() * F.pixelSize;
    float offy = rng.NextFloat() * F.pixelSize;
    float u = ((float)(int)i * F.pixelSize + F.originX + offx);
    float v = ((float)(int)j * F.pixelSize + F.originY + offy);
    v3 dir = mk3(F.front[0], F.front[1], F.front[2]) + mk3(F.right[0], F.right[1], F.right[2]) * u + mk3(F.up[0], F.up[1], F.up[2]) * v;
    d = normalize3(dir);
    o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
}

MIPT_DEV void add_sample(float* sample, uint32_t pix, v3 c) {
    float* p = sample + (size_t)pix * 4;
    if (c.x != 0.f) unsafeAtomicAdd(p + 0, c.x);
    if (c.y != 0.f) unsafeAtomicAdd(p + 1, c.y);
    if (c.z != 0.f) unsafeAtomicAdd(p + 2, c.z);
}

// =================================================================================================================
// Wavefront level kernel.  Persistent grid-stride over the input queue; every lane handles one ray per iteration.
// =================================================================================================================
constexpr uint32_t kLdsSceneVec = 192;     // float4 slots (3 KiB) for an LDS-resident scene (the Cornell box needs 168); with the 9-entry stack a 256-thread block takes exactly 40 KiB

template <bool PRIMARY, bool LDS_SCENE>
__global__ void __launch_bounds__(kBlock) k_level(LevelArgs A) {
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    __shared__ float4 ldsScene[LDS_SCENE ? kLdsSceneVec : 1];
    if (LDS_SCENE) {
        const uint32_t nn = A.S.nNodes * 4, nr = A.S.nRefs * 3;
        for (uint32_t i = threadIdx.x; i < nn; i += kBlock) ldsScene[i] = A.S.nodes[i];
        for (uint32_t i = threadIdx.x; i < nr; i += kBlock) ldsScene[nn + i] = A.S.refs[i];
        __syncthreads();
    }
    const uint32_t n = PRIMARY ? A.npix : *A.inCount;
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) {
        atomicAdd(A.stats, (unsigned long long)n);
        atomicAdd(A.stats + 1 + A.level, (unsigned long long)n);
    }
    const bool spawn = A.level + 1 < A.F.maxDepth;            // children have ttl-1 > 0 (CUDATracerProg.cu:433)
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += stride) {
        const uint32_t idx = base + threadIdx.x;
        const bool active = idx < n;
        v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), beta = mk3(1.f, 1.f, 1.f);
        uint32_t pix = 0, node = 1, s0 = 0;
        bool has1 = false, has2 = false;
        Bounce b;
        if (active) {
            if (PRIMARY) {
                pix = A.pix0 + idx;
                gen_primary(A.F, pix, o, d, s0);
            } else {
                float4 qa = A.qin.a[idx], qb = A.qin.b[idx], qc = A.qin.c[idx];
                o = xyz(qa); d = mk3(qa.w, qb.x, qb.y); beta = mk3(qb.z, qb.w, qc.x);
                pix = __float_as_uint(qc.y); node = __float_as_uint(qc.z); s0 = __float_as_uint(qc.w);
            }
            if (A.pixelRays) atomicAdd(A.pixelRays + pix, 1u);
            HitRec h = LDS_SCENE ? traverse((const float4*)ldsScene, (const float4*)ldsScene + A.S.nNodes * 4, o, d, ldsStack + threadIdx.x)
                       : A.S.wide ? traverse_wide(A.S.nodes, A.S.refs, o, d, ldsStack + threadIdx.x)
                                  : traverse(A.S.nodes, A.S.refs, o, d, ldsStack + threadIdx.x);
            if (h.ref < 0) {
                add_sample(A.sample, pix, beta * on_miss(A.S, d));
            } else {
                Surface sf = surface_at(A.S, o, d, h);
                Material mat = load_material(A.S, sf.mesh);
                v3 emit = mat.emissive * mat.emissiveStrength;
                add_sample(A.sample, pix, beta * emit);
                if (spawn) {
                    Lcg rng; rng.state = node_seed(s0, node);
                    b = shade_hit(mat, sf, d, h.front, rng);
                    b.f1 = beta * b.f1; b.f2 = beta * b.f2;
                    has1 = true; has2 = true;
                }
            }
        }
        if (spawn) {     // wave-uniform
            // wave64 compaction: ballot + prefix popcount, one atomic per wave (lane order = ray order is kept)
            unsigned long long m1 = __ballot(has1), m2 = __ballot(has2);
            uint32_t total = __popcll(m1) + __popcll(m2);
            if (total) {
                uint32_t lane = __lane_id();
                unsigned long long below = (1ull << lane) - 1ull;
                uint32_t off = __popcll(m1 & below) + __popcll(m2 & below);
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(A.outCount, total);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                uint32_t w = wbase + off;
                if (has1) {
                    A.qout.a[w] = make_float4(b.o1.x, b.o1.y, b.o1.z, b.d1.x);
                    A.qout.b[w] = make_float4(b.d1.y, b.d1.z, b.f1.x, b.f1.y);
                    A.qout.c[w] = make_float4(b.f1.z, __uint_as_float(pix), __uint_as_float(node * 2u), __uint_as_float(s0));
                    w++;
                }
                if (has2) {
                    A.qout.a[w] = make_float4(b.o2.x, b.o2.y, b.o2.z, b.d2.x);
                    A.qout.b[w] = make_float4(b.d2.y, b.d2.z, b.f2.x, b.f2.y);
                    A.qout.c[w] = make_float4(b.f2.z, __uint_as_float(pix), __uint_as_float(node * 2u + 1u), __uint_as_float(s0));
                }
            }
        }
    }
}

void launch_level(const LevelArgs& a, bool primary, int blocks, hipStream_t s) {
    const bool lds = !a.S.wide && a.S.nNodes * 4 + a.S.nRefs * 3 <= kLdsSceneVec;
    if (primary) { if (lds) hipLaunchKernelGGL((k_level<true, true>), dim3(blocks), dim3(kBlock), 0, s, a);
                   else hipLaunchKernelGGL((k_level<true, false>), dim3(blocks), dim3(kBlock), 0, s, a); }
    else { if (lds) hipLaunchKernelGGL((k_level<false, true>), dim3(blocks), dim3(kBlock), 0, s, a);
           else hipLaunchKernelGGL((k_level<false, false>), dim3(blocks), dim3(kBlock), 0, s, a); }
}

// =================================================================================================================
// Persistent wave-private work stacks with in-wave refill (the default integrator).
//
// The level-synchronous kernel above pays one returning atomic on ONE counter per wave-batch to append to the next
// level's queue; on MI355X a single word sustains only ~90 such atomics per microsecond, which bounded every
// non-final level (measured: profiles/r1).  Here nothing is shared between waves: a wave keeps its pending rays in
// its own LIFO region of HBM (hot top in L2) and pushes children with the same ballot/prefix compaction -- the
// "allocation" is a wave-uniform stack pointer.  One launch renders the whole frame.
//
// Lanes are kept busy in both halves of the work:
//  * traversal: a lane whose ray has terminated parks its (ray, hit) record in a per-wave LDS shade queue and goes
//    idle; once kRefill lanes are idle they are refilled together from the top of the wave's stack, or with new primary
//    rays claimed from one frame-wide pixel counter (one atomic per refill), so short rays do not wait for the longest
//    traversal in the wave;
//  * shading: runs only on full batches of 64 parked records (or on what is left when the wave is starving), so hit
//    and miss code never executes for a handful of lanes.
// Rays are independent and carry their own RNG key (node id), so the order they are processed in does not change any
// ray, only the order of the atomic adds into a pixel.
//
// Stack size: with strict 64-ray batches each recursion level holds <= 128 entries; out-of-order refill loosens that
// (simulated worst case, every ray spawning two children: ~160 per level), so the host sizes the stack at
// 256 * maxDepth + 256 entries per wave and the kernel raises an error flag instead of writing past it.
// =================================================================================================================
// shade-queue entries per wave (a terminated ray that finds it full parks one round later); sized so that a
// wave uses exactly a sixteenth of the CU's 160 KiB: cap x 64 B + 3 KiB traversal stack (LDS scenes: 2 KiB + a share of the 4 KiB scene); 12 stack entries / cap 112 measured better than 8 / 128 and 16 / 96
MIPT_DEV uint32_t band_pixel(const Band& b, uint32_t local) { return ((local / b.pixels) * b.stride + b.first) * b.pixels + local % b.pixels; }
// LDS-resident scenes are tiny trees: 8 LDS stack entries cover them (the host checks stackNeed), which leaves their
// blocks the same 112-entry queue next to the 4 KiB scene window (96 entries with a 12-entry stack: 1 % slower on Cornell)
constexpr int kLdsSceneStack = 8;       // the builder's bound an LDS-resident scene must meet; the array has one spare entry on top (see the BVH2 visit)
template <bool LDS_SCENE> struct SqCap { static constexpr int value = 112; };
constexpr int kSqWords = 16;         // dwords per entry, SoA by dword: conflict-free LDS access

// wave-stack records: plain cached accesses.  The stack is LIFO, so what was pushed last is popped next and L2 is exactly
// where it should sit; non-temporal stores / loads were measured 13 % (Cornell) and 9 % (Default) slower, grid unchanged
// (profiles/r2/exp_rejected_variants.log)
MIPT_DEV void stack_store(float4* p, float4 v) { *p = v; }
MIPT_DEV float4 stack_load(const float4* p) { return *p; }

template <bool LDS_SCENE>
MIPT_DEV void load_node(const float4* lds, const float4* glb, uint32_t cur, float4& n0, float4& n1, float4& n2, float4& n3) {
    if constexpr (LDS_SCENE) { const float4* N = lds + (size_t)cur * 4; n0 = N[0]; n1 = N[1]; n2 = N[2]; n3 = N[3]; }
    else {
        const float4* N = glb + (size_t)cur * 4;
        n0 = N[0]; n1 = N[1]; n2 = N[2]; n3 = N[3];
    }
}
template <bool LDS_SCENE>
MIPT_DEV void load_ref(const float4* lds, const float4* glb, uint32_t i, float4& a, float4& b, float4& c) {
    if constexpr (LDS_SCENE) { const float4* T = lds + (size_t)i * 3; a = T[0]; b = T[1]; c = T[2]; }
    else {
        const float4* T = glb + (size_t)i * 3;
        a = T[0]; b = T[1]; c = T[2];
    }
}

// ---- env-map importance sampling (option "estimator" = 1; specification: include/mipt.h, tables built in api.hip) ----
constexpr uint32_t kShadowBit = 0x80000000u;   // flags carried in the pixel word of a ray record (pixel index < 2^29)
constexpr uint32_t kNoEnvBit  = 0x40000000u;   //   occlusion ray of the estimator / diffuse child whose miss adds nothing
constexpr uint32_t kNoEmitBit = 0x20000000u;   //   diffuse child that skips the emission of the surface it lands on
constexpr uint32_t kPixMask   = 0x1FFFFFFFu;
MIPT_DEV uint32_t cdf_find(const float* __restrict__ cdf, uint32_t n, float u) {     // largest i in [0,n-1] with cdf[i] <= u
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid; else hi = mid; }
    return lo;
}
MIPT_DEV bool sample_env(const SceneDev& S, float u1, float u2, v3& dir, v3& L, float& wgt) {
    const uint32_t W = S.envW, H = S.envH;
    const float* row = S.envCdf;
    const uint32_t iv = cdf_find(row, H, u1);
    const float pr = row[iv + 1] - row[iv];
    const float* col = row + (H + 1) + (size_t)iv * (W + 1);
    const uint32_t iu = cdf_find(col, W, u2);
    const float pc = col[iu + 1] - col[iu];
    if (!(pr > 0.f) || !(pc > 0.f)) return false;
    const float du1 = (u1 - row[iv]) / pr, du2 = (u2 - col[iu]) / pc;
    const float theta = (((float)iv + du1) / (float)H) * 3.14159265358979323846f;
    const float phi = (((float)iu + du2) / (float)W) * 6.28318530717958647692f;      // azimuth + pi
    float st, ct, sp, cp; det_sincosf(theta, &st, &ct); det_sincosf(phi, &sp, &cp);
    dir = mk3(st * -sp, ct, st * -cp);
    const float* px = S.env + ((size_t)iv * W + iu) * 3;
    L = mk3(px[0], px[1], px[2]);
    wgt = st * S.neeScale / (pr * pc);
    return true;
}

// one point on the emissive triangles, uniform by area, as seen from x (surface normal N); false when it adds nothing.
// Lw = Le * p_cap / p_area * |cos_y| / dist^2, tmax = where the occlusion ray stops short of the emitter.
MIPT_DEV bool sample_emitter(const SceneDev& S, float r0, float r1, float r2, v3 x, v3 N, float minCos, v3& dir, v3& Lw, float& tmax) {
    const uint32_t k = cdf_find(S.emitCdf, S.nEmit, r0);
    const float* E = S.emitTri + (size_t)k * 15;
    const float su = __builtin_sqrtf(r1), b1 = 1 - su, b2 = r2 * su;
    v3 y = mk3(E[0], E[1], E[2]) + mk3(E[3], E[4], E[5]) * b1 + mk3(E[6], E[7], E[8]) * b2;
    v3 w = y - x;
    const float dist2 = dot3(w, w), dist = __builtin_sqrtf(dist2);
    if (!(dist > 0.f)) return false;
    dir = w / dist;
    if (!(dot3(dir, N) > minCos)) return false;
    const float cosy = __builtin_fabsf(dot3(mk3(E[9], E[10], E[11]), dir));
    if (!(cosy > 0.f)) return false;
    Lw = mk3(E[12], E[13], E[14]) * (cosy * S.emitScale / dist2);
    tmax = dist * 0.9999f;
    return true;
}

// Lane-utilisation instrumentation (make prof -> libmipt_prof.so, scripts/lane_profile.py).  Wave-level sums land in
// stats[8..]: [8] traversal rounds, [9] lanes active at round start, [10] node-loop wave iterations (max over lanes),
// [11] node visits (sum over lanes), [12] leaf-loop wave iterations, [13] triangle tests, [14] full shade batches,
// [15] lanes in them, [16] terminal batches, [17] lanes in them, [18] refills, [19] lanes refilled.
#ifdef MIPT_LANE_PROFILE
#define MIPT_PROF(x) x
MIPT_DEV uint32_t wave_max_u32(uint32_t v) { for (int s = 32; s > 0; s >>= 1) { uint32_t o = __shfl_xor(v, s); v = o > v ? o : v; } return v; }
MIPT_DEV uint32_t wave_sum_u32(uint32_t v) { for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s); return v; }
#else
#define MIPT_PROF(x)
#endif
// Phase clock (make variant NAME=phasetime EXTRA="-DMIPT_LANE_PROFILE -DMIPT_EXP_PHASETIME", scripts/exp_phase_time.py): the
// wave's s_memtime ticks since the previous mark go to phase k; the sums take the places of stats[8..13], [15] and [17] --
// 0 full shade batches, 1 terminal batches, 2 the end-of-frame check, 3 node loop, 4 leaf loop and the end-of-leaf rule, 5 parking,
// 6 the refill, 7 the loop top (batch decisions).
#if defined(MIPT_LANE_PROFILE) && defined(MIPT_EXP_PHASETIME)
#define MIPT_PT(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt_[k] += now_ - ptLast_; ptLast_ = now_; }
#else
#define MIPT_PT(k)
#endif

#define MIPT_BALLOT(x) __builtin_amdgcn_ballot_w64(x)      // wave64 mask of a per-lane predicate
// Threads per persistent block.  A block's CU slot is only handed to the next frame's blocks when all its waves have ended, so
// one-wave blocks let the next frame move in wave by wave: Default +3.9 %, grid +2.1 % (profiles/r2/exp_blocksize.log; 128
// threads: +0.5 %).  A one-wave block takes a sixteenth of the CU's LDS (3 KiB stack + 7 KiB shade queue).  LDS-resident
// scenes keep four waves per block: they share the block's copy of the scene.
template <bool LDS_SCENE> struct WaveStackBlock { static constexpr int value = LDS_SCENE ? 256 : 64; };
// 4 waves per SIMD: 103-113 VGPRs without light sampling since the device code is built without the SLP vectoriser (Makefile),
// 128 with it and for the light-sampling variants.  Measured before that: three waves with the ~145 VGPRs the compiler then
// wanted run 0.85x, five at 96 VGPRs spilt 35-40 registers, 0.6-0.7x; after it: five waves (96 VGPRs, 4-5 spills, 8 KiB of LDS
// per wave) Default +0.6 %, grid -4.5 % (profiles/r3/exp_codegen_flags.log).  Sixteen waves of 10 KiB fill the CU's LDS.
template <bool LDS_SCENE, bool WIDE, bool NEE, bool PBS = false, bool EMIT = false>      // EMIT: the scene has an emitters' tree ("emitters first", below)
__global__ void __launch_bounds__(WaveStackBlock<LDS_SCENE>::value, 4) k_wavestack(StackArgs A) {       // (hipcc: second argument = minimum waves per SIMD)
    // the body is wavestack_kernel.inc, once per residence of the scene (the reason is at its MIPT_BATCHES_BEGIN)
    if constexpr (LDS_SCENE)
#define MIPT_WS_LDS_SCENE 1
#include "wavestack_kernel.inc"
#undef MIPT_WS_LDS_SCENE
    else
#define MIPT_WS_LDS_SCENE 0
#include "wavestack_kernel.inc"
#undef MIPT_WS_LDS_SCENE
}

bool wavestack_scene_in_lds(const SceneDev& S) {
    return !S.wide && S.nNodes * 4 + S.nRefs * 3 <= kLdsSceneVec && S.stackNeed <= (uint32_t)kLdsSceneStack;
}

void launch_wavestack(const StackArgs& a, int blocks, hipStream_t s) {
    const bool lds = wavestack_scene_in_lds(a.S);
    if (a.nee == 3u) {      // estimator 3: the lobes of 2 with the light sampling of 1
        if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, true, true>), dim3(blocks * 4), dim3(64), 0, s, a);
        else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, true, true>), dim3(blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_wavestack<false, false, true, true>), dim3(blocks * 4), dim3(64), 0, s, a);
    }
    else if (a.nee == 2u) {      // estimator 2: VNDF specular + cosine diffuse, no occlusion rays
        if (a.S.emitRoot) {
            if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, false, true, true>), dim3(blocks * 4), dim3(64), 0, s, a);
            else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, false, true, true>), dim3(blocks), dim3(kBlock), 0, s, a);
            else hipLaunchKernelGGL((k_wavestack<false, false, false, true, true>), dim3(blocks * 4), dim3(64), 0, s, a);
        }
        else if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, false, true>), dim3(blocks * 4), dim3(64), 0, s, a);
        else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, false, true>), dim3(blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_wavestack<false, false, false, true>), dim3(blocks * 4), dim3(64), 0, s, a);
    }
    else if (a.nee) {
        if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, true>), dim3(blocks * 4), dim3(64), 0, s, a);
        else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, true>), dim3(blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_wavestack<false, false, true>), dim3(blocks * 4), dim3(64), 0, s, a);
    }
    else if (a.S.emitRoot) {     // the reference's estimator in a scene with an emitters' tree
        if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, false, false, true>), dim3(blocks * 4), dim3(64), 0, s, a);
        else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, false, false, true>), dim3(blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_wavestack<false, false, false, false, true>), dim3(blocks * 4), dim3(64), 0, s, a);
    }
    else if (a.S.wide) hipLaunchKernelGGL((k_wavestack<false, true, false>), dim3(blocks * 4), dim3(64), 0, s, a);
    else if (lds) hipLaunchKernelGGL((k_wavestack<true, false, false>), dim3(blocks), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((k_wavestack<false, false, false>), dim3(blocks * 4), dim3(64), 0, s, a);
}

// =================================================================================================================
// Depth-first integrator (rng_mode = dfs): the reference's evaluation order, bit for bit.
//
// One lane = one pixel-sample, like the reference's megakernel, with LaunchRayRecur's recursion
// (CUDATracerProg.cu:426-541) unrolled into an explicit per-lane frame stack.  The single LCG of the sample is
// consumed in the reference's depth-first order (GGX draws, the whole specular subtree, then the diffuse draws), and
// radiance is combined bottom-up with the reference's expressions, so the sample equals the oracle's DFS mode exactly.
// This is the fidelity path (divergent, scratch-heavy); the wave-stack integrator is the fast one.
// =================================================================================================================
constexpr int kDfsMaxDepth = 16;
struct DfsFrame {
    v3 P, N, d, R, color;
    float rr, rt, n1, n2;
    uint32_t mesh; int ttl; int stage;
};

__global__ void __launch_bounds__(kBlock) k_dfs(DfsArgs A) {
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= A.npix) return;
    const uint32_t pix = A.pix0 + idx;
    DfsFrame fr[kDfsMaxDepth];
    v3 o, d; uint32_t s0;
    Lcg rng; rng.state = tea16(pix, A.F.frameID);
    {   // primary ray: same draws as gen_primary, but the generator keeps running afterwards
        uint32_t i = pix % A.F.W, j = pix / A.F.W;
        float offx = rng.NextFloat() * A.F.pixelSize;
        float offy = rng.NextFloat() * A.F.pixelSize;
        float u = ((float)(int)i * A.F.pixelSize + A.F.originX + offx);
        float v = ((float)(int)j * A.F.pixelSize + A.F.originY + offy);
        v3 dir = mk3(A.F.front[0], A.F.front[1], A.F.front[2]) + mk3(A.F.right[0], A.F.right[1], A.F.right[2]) * u + mk3(A.F.up[0], A.F.up[1], A.F.up[2]) * v;
        d = normalize3(dir);
        o = mk3(A.F.camPos[0], A.F.camPos[1], A.F.camPos[2]);
        s0 = 0; (void)s0;
    }
    int depth = 0, ttl = (int)A.F.maxDepth;
    uint32_t traced = 0;
    v3 ret = mk3(0, 0, 0);
    bool entering = true;
    while (true) {
        if (entering) {
            // ---- LaunchRayRecur entry (:428-531)
            if (ttl <= 0) { ret = mk3(0, 0, 0); entering = false; continue; }
            traced++;
            HitRec h = A.S.wide ? traverse_wide(A.S.nodes, A.S.refs, o, d, ldsStack + threadIdx.x)
                                : traverse(A.S.nodes, A.S.refs, o, d, ldsStack + threadIdx.x);
            if (h.ref < 0) { ret = on_miss(A.S, d); entering = false; continue; }
            Surface sf = surface_at(A.S, o, d, h);
            Material mat = load_material(A.S, sf.mesh);
            v3 mN = ggx_rand_dir_inside_cone(rng, sf.N, mat.roughness);
            v3 R = reflect3(d, mN);
            float n1 = 1.f, n2 = mat.n; n2 += FLT_EPSILON;
            if (!h.front) { float t = n1; n1 = n2; n2 = t; }
            const float r0 = mat.r0;      // ((n1 - n2) / (n1 + n2))^2: the same for either order of n1, n2, so computed once per material on the host (api.hip fill_materials)
            float rr = (1 - dot3(-d, mN));
            rr = pow5_mul(rr);
            rr = r0 + (1 - r0) * rr;
            rr = saturatef(rr);
            rr = mat.metallic * 1.f + (1 - mat.metallic) * rr;
            DfsFrame& f = fr[depth];
            f.P = sf.P; f.N = sf.N; f.d = d; f.R = R; f.rr = rr; f.rt = 1 - rr; f.n1 = n1; f.n2 = n2; f.mesh = sf.mesh; f.ttl = ttl; f.stage = 1;
            o = sf.P + sf.N * 1e-6f; d = R;
            ttl -= 1; depth += 1;
            continue;
        }
        // ---- a call has returned `ret`
        if (depth == 0) break;
        depth -= 1;
        DfsFrame& f = fr[depth];
        Material mat = load_material(A.S, f.mesh);
        if (f.stage == 1) {
            v3 c = mk3(__builtin_fmaxf(ret.x, 0.f), __builtin_fmaxf(ret.y, 0.f), __builtin_fmaxf(ret.z, 0.f));   // :459
            v3 absorb = mat.metallic * mat.color + (1 - mat.metallic) * mk3(1.f, 1.f, 1.f);
            f.color = absorb * c * f.rr;
            if (mat.opacity < 1.f) {
                v3 frontv = (f.d + f.R) / 2.0f;
                v3 down = f.d - frontv;
                float down_len = length3(down);
                down = down_len < 1e-10f ? mk3(1, 0, 0) : (down / down_len);
                v3 front_refr = frontv * (f.n1 / f.n2);
                float frl = length3(front_refr);
                if (frl > 1) frl = 1;
                float drl = __builtin_sqrtf(1 - frl * frl);
                d = normalize3(front_refr + down * drl);
                o = f.P - f.N * 1e-6f;
            } else {
                d = rand_dir_inside_cone(rng, f.N, 0.49f);
                o = f.P + f.N * 1e-6f;
            }
            f.stage = 2;
            ttl = f.ttl - 1; depth += 1;
            entering = true;
        } else {
            v3 color = f.color + mat.color * ret * f.rt * ((mat.opacity == 1.f) ? 1.f : (1 - mat.opacity));   // :526
            color = color + mat.emissive * mat.emissiveStrength;                                                // :528
            ret = color;
        }
    }
    float* p = A.sample + (size_t)pix * 4;
    p[0] = ret.x; p[1] = ret.y; p[2] = ret.z;
    if (A.pixelRays) A.pixelRays[pix] = traced;
    for (int sh = 32; sh > 0; sh >>= 1) traced += __shfl_down(traced, sh);
    if ((threadIdx.x & 63) == 0 && traced) atomicAdd(A.stats, (unsigned long long)traced);
}
void launch_dfs(const DfsArgs& a, int blocks, hipStream_t s) {
    (void)blocks;
    if (!a.npix) return;
    hipLaunchKernelGGL(k_dfs, dim3((a.npix + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

// =================================================================================================================
// Accumulate + tonemap (tail of GenRay, CUDATracerProg.cu:892-919)
// =================================================================================================================
MIPT_DEV unsigned char tonemap1(float x) {
    x = saturatef(x);
    x = x / (x + 1.0f);
    x = det_pow_gamma(x);
    return (unsigned char)(int)(x * 255);
}
// zeroAfter: the sample slot is cleared as soon as it has been read, so the next frame that uses it needs no clear of
// its own (one 16 B/pixel fill kernel per frame less)
// zeroCounters: the 64 pixel counters of the frame being folded, cleared for the next frame that uses the block
// One-wave blocks: the fold runs next to the persistent integrator, whose slots free up one wave at a time; a 256-thread
// block would wait until four of them are free on one CU at once (grid: 8.64 -> 8.87 ms per frame when it did).
constexpr int kFoldBlock = 64;
__global__ void __launch_bounds__(kFoldBlock) k_accumulate(float4* sample, float4* acc, uchar4* ldr, float weight, int sumMode, int zeroAfter,
                                                        uint32_t* zeroCounters, Band band, uint32_t pix0, uint32_t npix) {
    uint32_t i = blockIdx.x * kFoldBlock + threadIdx.x;
    if (zeroCounters && i < 64u) zeroCounters[i] = 0u;
    if (i >= npix) return;
    uint32_t p = pix0 + i;
    if (band.pixels) p = band_pixel(band, p);
    float4 c = sample[p], a = acc[p];
    if (zeroAfter) sample[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    float r, g, b;
    if (sumMode) { r = a.x + c.x; g = a.y + c.y; b = a.z + c.z; }
    else { r = a.x * (1 - weight) + c.x * weight; g = a.y * (1 - weight) + c.y * weight; b = a.z * (1 - weight) + c.z * weight; }
    acc[p] = make_float4(r, g, b, 1.f);
    if (ldr) ldr[p] = make_uchar4(tonemap1(r), tonemap1(g), tonemap1(b), 255);
}
void launch_accumulate(float* sample, float* acc, unsigned char* ldr, float weight, int sumMode, int zeroAfter, uint32_t* zeroCounters, Band band, uint32_t pix0, uint32_t npix, hipStream_t s) {
    if (!npix) return;
    hipLaunchKernelGGL(k_accumulate, dim3((npix + kFoldBlock - 1) / kFoldBlock), dim3(kFoldBlock), 0, s, (float4*)sample, (float4*)acc, (uchar4*)ldr,
                       weight, sumMode, zeroAfter, zeroCounters, band, pix0, npix);
}
__global__ void __launch_bounds__(kFoldBlock) k_tonemap(const float4* acc, float scale, uint32_t npix, uchar4* ldr) {
    uint32_t i = blockIdx.x * kFoldBlock + threadIdx.x;
    if (i >= npix) return;
    float4 a = acc[i];
    ldr[i] = make_uchar4(tonemap1(a.x * scale), tonemap1(a.y * scale), tonemap1(a.z * scale), 255);
}
void launch_tonemap(const float* acc, float scale, uint32_t npix, unsigned char* ldr, hipStream_t s) {
    if (!npix) return;
    hipLaunchKernelGGL(k_tonemap, dim3((npix + kFoldBlock - 1) / kFoldBlock), dim3(kFoldBlock), 0, s, (const float4*)acc, scale, npix, (uchar4*)ldr);
}

// =================================================================================================================
// Closest hit for a caller-supplied ray batch (the traversal kernel on its own)
// =================================================================================================================
struct HitOut { float t, u, v; int32_t tri, mesh, front; };
__global__ void __launch_bounds__(kBlock) k_trace_rays(SceneDev S, const float* o, const float* d, uint32_t n, HitOut* out) {
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    v3 ro = mk3(o[(size_t)i * 3], o[(size_t)i * 3 + 1], o[(size_t)i * 3 + 2]);
    v3 rd = mk3(d[(size_t)i * 3], d[(size_t)i * 3 + 1], d[(size_t)i * 3 + 2]);
    HitRec h = S.wide ? traverse_wide(S.nodes, S.refs, ro, rd, ldsStack + threadIdx.x) : traverse(S.nodes, S.refs, ro, rd, ldsStack + threadIdx.x);
    HitOut r; r.t = h.t; r.u = h.u; r.v = h.v; r.front = h.front ? 1 : 0;
    if (h.ref >= 0) {
        r.tri = h.tri;
        r.mesh = (int32_t)__float_as_uint(S.refs[(size_t)h.ref * 3 + 1].w);
    } else { r.tri = -1; r.mesh = -1; }
    out[i] = r;
}
void launch_trace_rays(const SceneDev& S, const float* o, const float* d, uint32_t n, void* hits, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_trace_rays, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, S, o, d, n, (HitOut*)hits);
}

// =================================================================================================================
// Device math self-test
// =================================================================================================================
__global__ void k_test_math(int fn, const float* a, const float* b, uint32_t n, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c, r = 0.f;
    switch (fn) {
        case 0: det_sincosf(a[i], &s, &c); r = s; break;
        case 1: det_sincosf(a[i], &s, &c); r = c; break;
        case 2: r = det_atan2f(a[i], b[i]); break;
        case 3: r = det_acosf(a[i]); break;
        case 4: r = det_pow_gamma(a[i]); break;
        case 5: r = pow5_mul(a[i]); break;
        case 6: r = pow10_mul(a[i]); break;
    }
    out[i] = r;
}
void launch_test_math(int fn, const float* a, const float* b, uint32_t n, float* out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_test_math, dim3((n + 255) / 256), dim3(256), 0, s, fn, a, b, n, out);
}

}  // namespace mipt
