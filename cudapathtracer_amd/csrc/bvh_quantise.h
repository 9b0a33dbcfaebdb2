// bvh_quantise.h -- one quantised 4-wide node from up to four child boxes (layout: bvh_builder.h).  Shared by the host
// collapse (bvh_builder.cpp) and the GPU builder (gpu_bvh.hip): the same source, so both write the same bytes.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define MIPT_HD __host__ __device__
#else
#define MIPT_HD
#endif

namespace mipt {

struct QBox {
    float mn[3], mx[3];
    MIPT_HD bool valid() const { return mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]; }
    MIPT_HD float area() const {
        if (!valid()) return 0.f;
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx * dy + dy * dz + dz * dx) * 2.0f;
    }
};

MIPT_HD inline float q4_scale_of(uint32_t e) { uint32_t b = e << 23; float f; __builtin_memcpy(&f, &b, 4); return f; }

// writes one 4-wide node at P (16 floats) from up to four child boxes; false when an extent cannot be represented
MIPT_HD inline bool quantise_node(float* P, const QBox* ch, int n, const uint32_t* refs) {
    bool failed = false;
    QBox nb; for (int a = 0; a < 3; a++) { nb.mn[a] = INFINITY; nb.mx[a] = -INFINITY; }
    for (int i = 0; i < n; i++) if (ch[i].valid()) for (int a = 0; a < 3; a++) { nb.mn[a] = fminf(nb.mn[a], ch[i].mn[a]); nb.mx[a] = fmaxf(nb.mx[a], ch[i].mx[a]); }
    if (!nb.valid()) for (int a = 0; a < 3; a++) nb.mn[a] = nb.mx[a] = 0.f;
    uint32_t e[3]; uint8_t qlo[3][4], qhi[3][4];
    // a slot without a child gets an inverted box (entry plane behind the exit plane on every axis): the slab test
    // rejects it by itself (and should rounding ever let it through, its reference is a leaf of zero triangles)
    for (int a = 0; a < 3; a++) for (int i = 0; i < 4; i++) { qlo[a][i] = 255; qhi[a][i] = 0; }
    for (int a = 0; a < 3; a++) {
        const float origin = nb.mn[a];
        // smallest power of two with 255 * scale >= extent; kept inside [2^-60, 2^40] so the product with a ray's
        // inverse direction (1 <= |idir| <= 2^80) stays a normal, finite number
        int ex = 127 - 60;
        const float ext = nb.mx[a] - origin;
        while (ex < 167 && 255.0f * q4_scale_of((uint32_t)ex) < ext) ex++;
        for (;; ex++) {      // widen until every child plane fits (fp32 rounding of the extent can cost one step)
            bool ok = true;
            const float sc = q4_scale_of((uint32_t)ex);
            for (int i = 0; i < n && ok; i++) {
                if (!ch[i].valid()) { qlo[a][i] = 255; qhi[a][i] = 0; continue; }
                int lo = (int)floorf((ch[i].mn[a] - origin) / sc); lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
                while (lo > 0 && fmaf((float)lo, sc, origin) > ch[i].mn[a]) lo--;
                int hi = (int)ceilf((ch[i].mx[a] - origin) / sc); hi = hi < 0 ? 0 : hi;
                while (hi <= 255 && fmaf((float)hi, sc, origin) < ch[i].mx[a]) hi++;
                if (hi > 255 || fmaf((float)lo, sc, origin) > ch[i].mn[a]) { ok = false; break; }
                qlo[a][i] = (uint8_t)lo; qhi[a][i] = (uint8_t)hi;
            }
            if (ok) break;
            if (ex >= 167) { failed = true; break; }     // extent beyond 2^48: not representable
        }
        e[a] = (uint32_t)ex;
    }
    __builtin_memcpy(P, nb.mn, 12);
    // the per-axis scales as floats (powers of two): a visit multiplies them straight into the ray's inverse direction
    P[3] = q4_scale_of(e[2]);
    for (int a = 0; a < 3; a++) { __builtin_memcpy(P + 4 + a, qlo[a], 4); __builtin_memcpy(P + 7 + a, qhi[a], 4); }
    P[10] = q4_scale_of(e[0]); P[11] = q4_scale_of(e[1]);
    __builtin_memcpy(P + 12, refs, 16);
    return !failed;
}

}  // namespace mipt
