// kernels.h -- device-side data layout and launch wrappers shared by api.hip and kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

// Scene as the kernels see it.  Everything is flat, 16-byte aligned float4 streams in HBM:
//   nodes : 4 x float4 per BVH2 node (both children's boxes + two child refs)     64 B
//           or, when `wide` is set, per quantised 4-wide node (bvh_builder.h)     64 B
//   refs  : 3 x float4 per triangle reference (v0|triId, e1|meshId, e2|flags)      48 B   flags bit 1: the mesh emits
//   shade : 3 x float4 per triangle (unit vertex normals n0|meshId, n1, n2)        48 B, touched only for the winning hit
//   mats  : 3 x float4 per mesh (color|roughness, metallic opacity n emissiveStrength, emissiveColor|0)
struct SceneDev {
    const float4* nodes;
    const float4* refs;
    // the same tree for k_wavestack when it walks the scene from L2 / HBM: reference records stored in PAIRS (96 B: the two
    // triangles' components interleaved, v0x_a v0x_b v0y_a v0y_b | v0z_a v0z_b id_a id_b | e1x.. e1y.. | e1z_a e1z_b flags_a
    // flags_b | e2x.. e2y.. | e2z_a e2z_b 0 0) so that one leaf-loop iteration tests two triangles with packed fp32
    // arithmetic; every leaf starts a fresh pair (an odd leaf ends in a zero-area pad), and the leaf references of `nodes2`
    // hold the PAIR index in their low 27 bits (count unchanged).  Built by api.hip build_pairs from nodes / refs.
    const float4* nodes2;
    const float4* refs2;
    const float4* shade;
    const float4* mats;
    const float*  env;        // float RGB rows, or nullptr
    uint32_t envW, envH;
    float skyColor[3]; float skyDir[3]; float skyIntensity;
    uint32_t nNodes, nRefs;
    uint32_t wide;            // 0: BVH2 nodes, 1: quantised 4-wide nodes
    uint32_t stackNeed;       // the builder's bound of traversal-stack entries per ray (incl. the sentinel)
    const float* envCdf;      // estimator 1: rowCdf[envH+1], then envH rows of colCdf[envW+1]; nullptr without an env map
    float neeScale;           // pi / (0.98 * envW * envH)
    const float* emitTri;     // estimator 1: emissive triangles, 15 floats each [v0 e1 e2 unit-normal Le]; nullptr if none
    const float* emitCdf;     // area CDF, nEmit + 1 entries
    uint32_t nEmit; float emitScale;   // total emissive area / (2 pi 0.98)
    uint32_t emissiveMeshes;  // meshes whose hits can add emission (strength != 0 and a non-black colour); 0: a hit at the last
                              // level adds nothing, whichever triangle it is, so those rays only need to know hit-or-miss
    // Emitters first (scenes with emitting meshes): behind the main tree's nodes and references sits a second, small tree over
    // the emitting triangles alone.  emitRoot = its root reference (an inner node index >= the main tree's node count, or a
    // leaf reference when the emitters fit one leaf); 0 = none.  References from nRefsMain on belong to it.  (k_wavestack, D)
    uint32_t emitRoot, nRefsMain;
};

// Per-frame constants computed once on the host (CUDATracerProg.cu:369-397 computes them per thread).
struct FrameDev {
    float camPos[3], front[3], right[3], up[3];
    float pixelSize, originX, originY;
    uint32_t W, H, frameID, maxDepth;
};

// Ray queue, structure of arrays of float4 so that a wave reads/writes 1 KiB per instruction:
//   a = (o.x o.y o.z d.x)  b = (d.y d.z beta.r beta.g)  c = (beta.b, pixel, node, s0)
struct Queue { float4* a; float4* b; float4* c; };

struct LevelArgs {
    SceneDev S; FrameDev F;
    Queue qin, qout;
    const uint32_t* inCount;     // rays in qin (ignored for the primary level)
    uint32_t* outCount;          // rays appended to qout
    float* sample;               // per-pixel radiance of this frame, 4 floats per pixel (rgb + unused)
    uint32_t* pixelRays;         // optional per-pixel ray counter (tests), may be nullptr
    unsigned long long* stats;   // [0] rays traced, [1+level] rays traced at that level
    uint32_t level;              // depth of the rays in qin (0 = primary)
    uint32_t pix0, npix;         // pixel tile handled by this pass
};

// Persistent wave-private work stacks (the default integrator): every wave owns `cap` ray slots of the three SoA
// arrays and runs its rays' whole recursion trees out of that LIFO, refilling idle lanes with fresh primary rays.
// Row-tile partition of the frame (mipt_set_tile_partition): tile k = `pixels` consecutive pixels (whole rows); this
// handle renders tiles first, first + stride, ...  pixels == 0: the whole frame.  Local pixel index L of the partition
// maps to frame pixel ((L / pixels) * stride + first) * pixels + L % pixels.
struct Band { uint32_t pixels, stride, first; };

struct StackArgs {
    SceneDev S; FrameDev F;
    Queue stack;                 // waves * cap entries, wave w owns [w*cap, (w+1)*cap)
    uint32_t refill;             // refill idle lanes once this many are idle (1..64)
    uint32_t minDescend;         // a traversal round stops descending when fewer lanes than this are still at inner nodes
    uint32_t nee;                // option "estimator": 0 the reference's, 1 light sampling (occlusion rays), 2 VNDF + cosine lobes
    uint32_t cap;                // entries per wave
    uint32_t* pixCounter;        // next unclaimed pixel of the frame (reset to 0 before the launch); with xcdBands: 8
                                 // counters, 8 words apart, one per band
    uint32_t xcdBands;           // 0: one pixel counter.  1: the frame is cut into 8 bands of consecutive pixels and a block
                                 // claims from band blockIdx % 8 first (blocks b and b + 8 share an XCD and its L2), then
                                 // from the bands after it until all are dry
    float* sample; uint32_t* pixelRays; unsigned long long* stats;
    uint32_t pix0, npix;         // npix: pixels of this handle's partition (the whole frame without one)
    Band band;
    uint32_t alphaOne;           // 1: `sample` is the caller's accumulation buffer (sum mode): every primary ray also writes alpha = 1
};

struct DfsArgs {
    SceneDev S; FrameDev F;
    float* sample; uint32_t* pixelRays; unsigned long long* stats;
    uint32_t pix0, npix;
};

void launch_level(const LevelArgs& a, bool primary, int blocks, hipStream_t s);
void launch_wavestack(const StackArgs& a, int blocks, hipStream_t s);
bool wavestack_scene_in_lds(const SceneDev& S);    // true: k_wavestack walks this scene from LDS (BVH2, fits the window and the LDS stack)
void launch_dfs(const DfsArgs& a, int blocks, hipStream_t s);
void launch_accumulate(float* sample, float* acc, unsigned char* ldr, float weight, int sumMode, int zeroAfter, uint32_t* zeroCounters, Band band,
                       uint32_t pix0, uint32_t npix, hipStream_t s);
void launch_tonemap(const float* acc, float scale, uint32_t npix, unsigned char* ldr, hipStream_t s);
void launch_trace_rays(const SceneDev& S, const float* o, const float* d, uint32_t n, void* hits, hipStream_t s);
void launch_test_issue(int packed, uint32_t iters, int blocks, unsigned long long* stamps, float* sink, hipStream_t s);   // microbench.hip
void launch_test_rcp_sweep(int steps, int blocks, unsigned long long* out, hipStream_t s);      // microbench.hip
void launch_test_math(int fn, const float* a, const float* b, uint32_t n, float* out, hipStream_t s);

}  // namespace mipt
