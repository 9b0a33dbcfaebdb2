// microbench.hip -- calibration kernels for the roofline bench.py reports (no counterpart in the reference).
//
// k_test_issue settles what one SIMD of gfx950 can issue: `iters` x 64 independent v_fma_f32 per wave (8 accumulator
// chains, so no instruction waits for the one before it), launched on the integrator's own grid shape
// (256-thread blocks, numCU x blocksPerCU blocks: one wave per SIMD per block, blocksPerCU waves per SIMD).
// Each wave stamps the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) around its loop and lane 0
// stores them with plain vector stores, so the host gets wave-instructions per SIMD-cycle from the kernel's own clock
// (no assumption about the clock the chip holds under load) next to the HIP-event time of the launch.
// MI355X_MICROARCH.md states 2 cycles per wave64 VALU instruction with several waves per SIMD and 4 for a wave alone;
// scripts/issue_calibration.py records what this kernel measures, with the SQ counters of the same launches.
#include "kernels.h"
#include "device_math.h"

namespace mipt {

#define MIPT_FMA8_TXT                     \
    "v_fma_f32 %0, %8, %9, %0\n\t"        \
    "v_fma_f32 %1, %8, %9, %1\n\t"        \
    "v_fma_f32 %2, %8, %9, %2\n\t"        \
    "v_fma_f32 %3, %8, %9, %3\n\t"        \
    "v_fma_f32 %4, %8, %9, %4\n\t"        \
    "v_fma_f32 %5, %8, %9, %5\n\t"        \
    "v_fma_f32 %6, %8, %9, %6\n\t"        \
    "v_fma_f32 %7, %8, %9, %7\n\t"
// one asm statement for all 64: between separate asm statements the compiler pads with s_nop
#define MIPT_FMA64(x_, y_)                                                                                            \
    asm volatile(MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT MIPT_FMA8_TXT \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x_), "v"(y_))

// the packed form: v_pk_fma_f32 does two fp32 FMAs per lane; the chip's 157.3 TFLOP/s fp32 vector peak (spec) is quoted on it
#define MIPT_PKFMA8_TXT                             \
    "v_pk_fma_f32 %0, %8, %9, %0\n\t"               \
    "v_pk_fma_f32 %1, %8, %9, %1\n\t"               \
    "v_pk_fma_f32 %2, %8, %9, %2\n\t"               \
    "v_pk_fma_f32 %3, %8, %9, %3\n\t"               \
    "v_pk_fma_f32 %4, %8, %9, %4\n\t"               \
    "v_pk_fma_f32 %5, %8, %9, %5\n\t"               \
    "v_pk_fma_f32 %6, %8, %9, %6\n\t"               \
    "v_pk_fma_f32 %7, %8, %9, %7\n\t"
#define MIPT_PKFMA64(x_, y_)                                                                                                      \
    asm volatile(MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT MIPT_PKFMA8_TXT \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(x_), "v"(y_))

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_test_issue_pk(uint32_t iters, float seed, unsigned long long* stamps, float* sink) {
    float2v p0 = { seed + (float)threadIdx.x, seed }, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    const float2v x = { 0.999f + seed, 0.998f + seed }, y = { 1e-3f * seed, 2e-3f * seed };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < iters; i++) {
        MIPT_PKFMA64(x, y);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0u) { stamps[2 * (size_t)wave] = c1 - c0; stamps[2 * (size_t)wave + 1] = r1 - r0; }
    const float2v s2 = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
    if (s2.x + s2.y == 12345.678f) sink[threadIdx.x] = s2.x;
}

__global__ void __launch_bounds__(256) k_test_issue(uint32_t iters, float seed, unsigned long long* stamps, float* sink) {
    float a0 = seed + (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float x = 0.999f + seed, y = 1e-3f * seed;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < iters; i++) {
        MIPT_FMA64(x, y);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0u) { stamps[2 * (size_t)wave] = c1 - c0; stamps[2 * (size_t)wave + 1] = r1 - r0; }
    const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 12345.678f) sink[threadIdx.x] = s;      // keeps the chains live; practically never taken
}

// Exhaustive check of rcp_newton<STEPS> against 1.0f / x: every float whose magnitude lies in [2^-100, 2^100], both signs.
// out[0] = values compared, out[1] = values where the bits differ, out[2] = the first such input found (bits), out[3] = how
// many differ by more than one unit in the last place.
template <int STEPS>
__global__ void __launch_bounds__(256) k_test_rcp_sweep(unsigned long long* out) {
    unsigned long long n = 0, bad = 0, worse = 0;
    const uint32_t stride = gridDim.x * 256u;
    for (unsigned long long p = blockIdx.x * 256u + threadIdx.x; p < (1ull << 32); p += stride) {
        const uint32_t bits = (uint32_t)p, ex = (bits >> 23) & 0xFFu;
        if (ex < 27u || ex > 227u) continue;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x, got = rcp_newton<STEPS>(x);
        n++;
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            const int du = (int)__float_as_uint(want) - (int)__float_as_uint(got);
            if (du > 1 || du < -1) worse++;
            atomicMin(out + 2, (unsigned long long)bits | (1ull << 40));
        }
    }
    atomicAdd(out + 0, n); if (bad) atomicAdd(out + 1, bad); if (worse) atomicAdd(out + 3, worse);
}
// The same for sqrt_unit against sqrtf: zero and every positive float in [2^-96, 2^100].
__global__ void __launch_bounds__(256) k_test_sqrt_sweep(unsigned long long* out) {
    unsigned long long n = 0, bad = 0, worse = 0;
    const uint32_t stride = gridDim.x * 256u;
    for (unsigned long long p = blockIdx.x * 256u + threadIdx.x; p < (1ull << 31); p += stride) {
        const uint32_t bits = (uint32_t)p, ex = (bits >> 23) & 0xFFu;
        if (bits != 0u && (ex < 31u || ex > 227u)) continue;
        const float x = __uint_as_float(bits);
        const float want = __builtin_sqrtf(x), got = sqrt_unit(x);
        n++;
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            const int du = (int)__float_as_uint(want) - (int)__float_as_uint(got);
            if (du > 1 || du < -1) worse++;
            atomicMin(out + 2, (unsigned long long)bits | (1ull << 40));
        }
    }
    atomicAdd(out + 0, n); if (bad) atomicAdd(out + 1, bad); if (worse) atomicAdd(out + 3, worse);
}
void launch_test_rcp_sweep(int steps, int blocks, unsigned long long* out, hipStream_t s) {
    if (steps == 4) { hipLaunchKernelGGL(k_test_sqrt_sweep, dim3(blocks), dim3(256), 0, s, out); return; }
    if (steps == 0) hipLaunchKernelGGL(k_test_rcp_sweep<0>, dim3(blocks), dim3(256), 0, s, out);      // the bare estimate: shows the sweep can see a difference
    else if (steps == 1) hipLaunchKernelGGL(k_test_rcp_sweep<1>, dim3(blocks), dim3(256), 0, s, out);
    else if (steps == 2) hipLaunchKernelGGL(k_test_rcp_sweep<2>, dim3(blocks), dim3(256), 0, s, out);
    else hipLaunchKernelGGL(k_test_rcp_sweep<3>, dim3(blocks), dim3(256), 0, s, out);
}

void launch_test_issue(int packed, uint32_t iters, int blocks, unsigned long long* stamps, float* sink, hipStream_t s) {
    if (packed) hipLaunchKernelGGL(k_test_issue_pk, dim3(blocks), dim3(256), 0, s, iters, 1.0f, stamps, sink);
    else hipLaunchKernelGGL(k_test_issue, dim3(blocks), dim3(256), 0, s, iters, 1.0f, stamps, sink);
}

}  // namespace mipt
