// device_math.h -- fp32 vector helpers, RNG and portable transcendentals for the gfx950 kernels.
//
// Arithmetic contract: IEEE fp32 evaluated exactly as written (the library is built with -ffp-contract=off and
// correctly rounded divide/sqrt), FMA only where fmaf is spelled out.  The CPU oracle states the same contract, so
// hit records, ray directions and ray counts can be compared bit for bit.
// Reference lines restated here: CUDATracer/CUDATracerProg/CUDATracerProg.cu:28-76 (LCG/TEA),
// CUDATracer/Math/Vec.hpp:306-348 (cross, dot, normalize, length, reflect).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

#define MIPT_DEV __device__ __forceinline__

struct v3 { float x, y, z; };
MIPT_DEV v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
MIPT_DEV v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MIPT_DEV v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MIPT_DEV v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
MIPT_DEV v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MIPT_DEV v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
MIPT_DEV v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
MIPT_DEV v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
MIPT_DEV float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MIPT_DEV float dot3(v3 a, v3 b) { return ffma(a.z, b.z, ffma(a.y, b.y, a.x * b.x)); }
MIPT_DEV v3 cross3(v3 a, v3 b) {
    return mk3(ffma(a.y, b.z, -(b.y * a.z)), ffma(a.z, b.x, -(b.z * a.x)), ffma(a.x, b.y, -(b.x * a.y)));
}
MIPT_DEV float length3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
MIPT_DEV v3 normalize3(v3 a) { return (a * 1.f) / __builtin_sqrtf(dot3(a, a)); }
MIPT_DEV v3 reflect3(v3 i, v3 n) { return i - n * dot3(n, i) * 2.0f; }
// 1 / x, correctly rounded, for |x| in [2^-100, 2^100] (no scaling, no special cases): the hardware estimate (1 ulp) and
// `steps` Newton corrections in FMA form.  mipt_test_rcp_sweep compares it with the compiler's full IEEE expansion of
// 1.0f / x (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup: 11 instructions) over every float in that range.
template <int STEPS>
MIPT_DEV float rcp_newton(float x) {
    float r = __builtin_amdgcn_rcpf(x);
#pragma unroll
    for (int i = 0; i < STEPS; i++) { const float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return r;
}
// sqrt(x), correctly rounded, for x = 0 or x in [2^-96, 2^100]: the compiler's IEEE expansion without the two rescalings that
// only inputs below 2^-96 need (v_sqrt_f32 is good to 1 ulp; the residuals of its two neighbours pick the rounded value) --
// 12 instructions instead of 17.  For the integrator's own unit-range arguments (a 24-bit uniform draw v, 1 - v, 1 - x * x
// with |x| <= 1: zero or at least 2^-24).  mipt_test_rcp_sweep(steps = 4) compares it with sqrtf over every float in the range.
MIPT_DEV float sqrt_unit(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float rlo = __builtin_fmaf(-lo, s, x), rhi = __builtin_fmaf(-hi, s, x);
    float r = rlo <= 0.f ? lo : s;
    r = rhi > 0.f ? hi : r;
    return x == 0.f ? x : r;
}
MIPT_DEV float saturatef(float x) { return !(x > 0.f) ? 0.f : (x > 1.f ? 1.f : x); }
MIPT_DEV v3 xyz(float4 f) { return mk3(f.x, f.y, f.z); }

// ---- RNG ---------------------------------------------------------------------------------------------------
MIPT_DEV uint32_t tea16(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (unsigned n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
struct Lcg {
    uint32_t state;
    MIPT_DEV float NextFloat() {
        state = 1664525u * state + 1013904223u;
        return (float)(state & 0x00FFFFFFu) / (float)0x01000000;
    }
};
MIPT_DEV uint32_t node_seed(uint32_t s0, uint32_t node) {
    uint32_t x = s0 ^ (node * 0x9E3779B9u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// ---- portable transcendentals (same polynomials, same operation order as the oracle's ORC_MATH_DET table) ---
MIPT_DEV void det_sincosf(float x, float* s, float* c) {
    float kf = __builtin_floorf(ffma(x, 0.636619772367581343f, 0.5f));
    int k = (int)kf;
    float r = ffma(-kf, 1.5703125f, x);
    r = ffma(-kf, 4.837512969970703125e-4f, r);
    r = ffma(-kf, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = ffma(ffma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = ffma(ps * z, r, r);
    float pc = ffma(ffma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = ffma(pc, z * z, ffma(-0.5f, z, 1.0f));
    int q = k & 3;
    float a = (q & 1) ? cs : sn;          // sin result before sign
    float b = (q & 1) ? sn : cs;          // cos result before sign
    *s = (q & 2) ? -a : a;
    *c = (q == 1 || q == 2) ? -b : b;
}
MIPT_DEV float det_atan_pos(float t) {
    // the three ranges share one division (t / 1 is t exactly): in a wave whose lanes fall into different ranges the two
    // divisions of the branchy form were issued one after the other
    const bool hi = t > 2.414213562373095f, mid = !hi && t > 0.4142135623730950f;
    const float y0 = hi ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.f);
    const float num = hi ? -1.0f : (mid ? t - 1.0f : t), den = hi ? t : (mid ? t + 1.0f : 1.0f);
    t = num / den;
    float z = t * t;
    float p = ffma(ffma(ffma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + ffma(p * z, t, t);
}
MIPT_DEV float det_atan2f(float y, float x) {
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a;
    if (ax == 0.f) a = (ay == 0.f) ? 0.f : 1.5707963267948966f;
    else a = det_atan_pos(ay / ax);
    if (x < 0.f) a = 3.14159265358979323846f - a;
    return (y < 0.f) ? -a : a;
}
MIPT_DEV float det_asin_poly(float x) {
    float z = x * x;
    float p = ffma(ffma(ffma(ffma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                   1.6666752422e-1f);
    return ffma(p * z, x, x);
}
MIPT_DEV float det_acosf(float x) {
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asin_poly(__builtin_sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asin_poly(__builtin_sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - det_asin_poly(x);
}
MIPT_DEV float det_logf(float x) {
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xff) - 126;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);
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
MIPT_DEV float det_expf(float x) {
    float nf = __builtin_floorf(ffma(x, 1.44269504088896341f, 0.5f));
    int n = (int)nf;
    float r = ffma(-nf, 0.693359375f, x);
    r = ffma(nf, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = ffma(p, r, 1.3981999507e-3f); p = ffma(p, r, 8.3334519073e-3f); p = ffma(p, r, 4.1665795894e-2f);
    p = ffma(p, r, 1.6666665459e-1f); p = ffma(p, r, 5.0000001201e-1f);
    float y = ffma(p, z, r) + 1.0f;
    float sc = __uint_as_float((uint32_t)(n + 127) << 23);
    return y * sc;
}
MIPT_DEV float det_pow_gamma(float x) {
    if (!(x > 1e-30f)) return 0.f;
    return det_expf((1.0f / 2.2f) * det_logf(x));
}
MIPT_DEV float pow5_mul(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
MIPT_DEV float pow10_mul(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; return x8 * x2; }

}  // namespace mipt
