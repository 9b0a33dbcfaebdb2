// api.hip -- host side of libmipt: the C-ABI declared in include/mipt.h.
//
// Owns all HIP state (one device + one stream per tracer handle), flattens scenes into the traversal layout and
// drives the wavefront integrator.  Replaces, for the hot path only: CUDATracerProg::Trace
// (CUDATracer/CUDATracerProg/CUDATracerProg.cu:937-963), CUDAScene::CUDAScene (CUDATracer/CUDATracer.hpp:133-203)
// and CUDABuffer (CUDATracer/CUDABuffer.cpp:111-224).
#include "../../include/mipt.h"
#include "bvh_builder.h"
#include "gpu_bvh.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>

using namespace mipt;

struct mipt_tracer {
    int device = 0;
    int numCU = 256;
    hipStream_t own = nullptr, stream = nullptr;
    std::string err;
    // options
    int64_t tilePixels = 0;          // 0 = derive from queueBudgetMB
    int64_t queueBudgetMB = 6144;
    int64_t rngMode = MIPT_RNG_NODE;
    int64_t countPixelRays = 0;
    int64_t blocksPerCU = 4;         // resident 256-thread blocks per CU of the persistent kernels
    int64_t refillLanes = 8;
    int64_t minDescend = 0;
    int64_t xcdBands = -1;             // -1 auto, 0 one pixel counter, 1 one band of the frame per XCD
    int64_t estimator = 0;
    int64_t profileLevels = 0;
    int64_t integrator = 0;          // 0 persistent wave stacks, 1 level-synchronous queues
    uint32_t tileRows = 0, tileParts = 1, tilePart = 0;     // mipt_set_tile_partition (tileParts 1: the whole frame)
    int64_t pipelineFrames = 1;      // consecutive frames alternate between two streams / buffer sets (see mipt_render)
    int64_t keepSample = 0;          // 1: a pipelined frame's sample buffer survives its fold (mipt_last_sample), at the price of one clear per frame
    int64_t waveStackEntries = 0;    // test option "wave_stack_entries": entries per wave the kernel may use (0 = 256 * maxDepth + 256)
    size_t guardFrom = 0;            // with that option: first entry of the wave-stack arrays no wave may touch (mipt_test_stack_guard)
    // work buffers
    Queue wstack = {}; size_t wstackEntries = 0;
    Queue q[2] = {};
    size_t qcap[2] = { 0, 0 };
    uint32_t* counts = nullptr;               // per-level queue counters
    unsigned long long* stats = nullptr;      // [0] rays
    float* sample = nullptr; size_t samplePixels = 0;
    uint32_t* pixelRays = nullptr; size_t pixelRaysCap = 0;
    // frame pipeline (mipt_render): set 1 of the integrator's buffers on `aux` (set 0 = the members above on `stream`),
    // a ring of sample buffers, and a high-priority stream for the folds into the caller's buffers
    hipStream_t aux = nullptr, fold = nullptr;
    Queue wstack2 = {}; size_t wstack2Entries = 0;
    uint32_t* counts2 = nullptr;
    static constexpr int kRing = 4;
    uint32_t* ringCounts = nullptr;          // kRing x 64 pixel counters of pipelined frames: the fold of a frame clears the block it used
    bool ringCountsClean[kRing] = {};
    float* ring[kRing] = {}; size_t ringPixels = 0;          // ring[0] aliases `sample`
    hipEvent_t kernEv[kRing] = {}, foldEv[kRing] = {};       // integrator of / fold of the frame that last used the slot
    bool foldValid[kRing] = {};
    size_t slotDirty[kRing] = {};    // pixels of the slot (from its start) that hold a sample no fold has cleared: the next frame using it clears them first
    bool sampleConsumed = false;     // the most recent frame left no sample behind (fold cleared it / summed straight into acc)
    std::vector<hipEvent_t> freeEvents;      // timing events are pooled: created at the high-water mark, reused, destroyed with the tracer
    // Sum mode without an LDR target: the integrator adds every ray's contribution straight into a library-owned block
    // buffer, which is folded into the caller's running sum every kBlockFrames frames (and whenever anything else looks at
    // the tracer: join_aux).  Adding straight into the caller's sum -- round 2 -- drops every contribution below half an
    // ulp of a sum that grows with the frame count: measured -3.7e-5 of the image mean after 2048 frames of the Default
    // scene under the HDRI, -1.1e-4 after 8192 (profiles/r3/exp_direct_sum.log).
    static constexpr uint32_t kBlockFrames = 64;
    float* blockSum = nullptr; size_t blockPixels = 0;      // npix x float4, zero outside an open block
    float* blockAcc = nullptr;                               // the caller's sum buffer the open block belongs to
    uint32_t blockFrames = 0, blockNpixPart = 0; Band blockBand{ 0u, 1u, 0u };
    bool blockPipe = false;                                  // the open block's frames ran on both pipeline streams
    hipEvent_t blockEvA = nullptr, blockEvB = nullptr, blockEvF = nullptr;
    void* staticScratch = nullptr;   // this device's copy of the file-scope `scratch` array (mipt_trace_static)
    hipEvent_t joinEv = nullptr, joinEv2 = nullptr;
    bool pipeBusy = false;           // frames are in flight that `stream` has not been ordered after yet
    bool mainTouched = true;         // something other than a pipelined frame was enqueued on `stream` since aux/fold last waited on it
    uint32_t flip = 0;               // frames rendered through the pipeline
    int lastSlot = 0;                // ring slot of the most recent frame (mipt_last_sample)
    // counters
    uint64_t frames = 0, launches = 0; double kernelMs = 0, kernelSpanMs = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    struct LevelEv { hipEvent_t a, b; int level; };   // level -1 = accumulate
    std::vector<LevelEv> levelPending;
    double levelMs[32] = {}; uint64_t levelLaunches[32] = {}; double accumulateMs = 0;
};

struct mipt_scene {
    mipt_tracer* t = nullptr;
    SceneDev dev = {};
    void* dNodes2 = nullptr; void* dRefs2 = nullptr; size_t nodes2Bytes = 0, refs2Bytes = 0;     // pair layout for k_wavestack (kernels.h)
    void* dNodes = nullptr; void* dRefs = nullptr; void* dShade = nullptr; void* dMats = nullptr; void* dEnv = nullptr; void* dEnvCdf = nullptr;
    std::vector<float> hEnvCdf;
    void* dEmitTri = nullptr; void* dEmitCdf = nullptr;
    std::vector<float> hEmitTri, hEmitCdf;      // estimator 1: emissive triangles (15 floats each) and their area CDF
    std::vector<float> hNodes, hRefs;    // host copies for mipt_scene_get_bvh
    std::vector<uint32_t> hRefTri;       // triangle id per reference (mipt_scene_refit)
    // "emitters first" (kernels.h SceneDev::emitRoot): a second tree over the emitting triangles alone, appended to the device's
    // node / reference arrays behind the main tree.  Host side: the tree with its own indices, the triangles it is built over
    // (global ids), its reference records; on the device: room for emitNodeCap nodes and emitRefCap references behind the main tree
    BvhOut hEmitBvh; std::vector<uint32_t> hEmitIds; std::vector<float> hEmitNodes, hEmitRefs; uint32_t emitRootRef = 0, emitNodeCap = 0, emitRefCap = 0, mainStackNeed = 0;
    BvhParams bvhParams;                         // what the scene was built with (the emitters' tree is rebuilt with it when materials change)
    uint32_t nMeshes = 0; size_t emitCap = 0;   // emitCap: emissive triangles the device buffers hold
    uint32_t nTris = 0;
    uint32_t bvhWidth = 2;
    bool bvhFromCache = false;
};

struct mipt_buffer {
    mipt_tracer* t = nullptr;
    void* cpu = nullptr; void* gpu = nullptr; size_t size = 0; int head = MIPT_HEAD_UNINITIALIZED;
    bool ownCpu = false, ownGpu = false;      // false: memory handed in through mipt_buffer_set_*_data (never freed here)
};

// ---- scratch of the allocation-free entry point (mipt_trace_static) ------------------------------------------------
// One file-scope device array per GPU the library is loaded on, held until the process ends.  Host code never uses
// the array's name as an address: mipt_create asks the runtime for this device's copy (hipGetSymbolAddress) and keeps
// the pointer in the tracer; kernels, memsets and copies take that pointer.
//   [0, 256)                      64 pixel counters
//   [256, 256 + 1 MiB)            sample buffer, kStaticMaxPixels x float4
//   then three SoA wave-stack arrays of kStaticBlocks x 4 waves x cap entries x 16 B (cap = 256 * maxDepth + 256)
constexpr size_t   kStaticScratchBytes = 32u << 20;
constexpr uint32_t kStaticMaxPixels = 1u << 16;         // e.g. 256 x 256
constexpr uint32_t kStaticMaxDepth = 8;
constexpr int      kStaticBlocks = 64;                  // 256 waves
constexpr size_t   kStaticSampleOffset = 256;
constexpr size_t   kStaticStackOffset = kStaticSampleOffset + (size_t)kStaticMaxPixels * 16;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static_assert(kStaticStackOffset + 3 * (size_t)kStaticBlocks * 4 * (256 * kStaticMaxDepth + 256) * 16 <= kStaticScratchBytes, "static scratch layout");

namespace {

int fail(mipt_tracer* t, int code, const std::string& msg) { if (t) t->err = msg; return code; }
#define HIPCHK(t, call)                                                                                   \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((t), MIPT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

int use_device(mipt_tracer* t) { hipError_t e = hipSetDevice(t->device); if (e != hipSuccess) return fail(t, MIPT_E_HIP, "hipSetDevice failed"); return MIPT_OK; }
// Orders `stream` after everything the frame pipeline has put on `aux`, so that whatever the caller does next on the
// tracer's stream (read-back, tonemap, synchronize ...) sees every frame rendered so far.
// Folds the open block of direct sum-mode frames into the caller's sum buffer (acc += block, block = 0) behind the frames
// that wrote it, and orders the streams those frames ran on after the fold, so the next block starts from zero.
void flush_block(mipt_tracer* t) {
    if (!t->blockFrames || !t->blockSum || !t->blockAcc) { t->blockFrames = 0; return; }
    if (t->blockPipe) {
        hipEventRecord(t->blockEvA, t->stream); hipEventRecord(t->blockEvB, t->aux);
        hipStreamWaitEvent(t->fold, t->blockEvA, 0); hipStreamWaitEvent(t->fold, t->blockEvB, 0);
        launch_accumulate(t->blockSum, t->blockAcc, nullptr, 1.f, 1, 1, nullptr, t->blockBand, 0, t->blockNpixPart, t->fold);
        hipEventRecord(t->blockEvF, t->fold);
        hipStreamWaitEvent(t->stream, t->blockEvF, 0); hipStreamWaitEvent(t->aux, t->blockEvF, 0);
    } else {
        launch_accumulate(t->blockSum, t->blockAcc, nullptr, 1.f, 1, 1, nullptr, t->blockBand, 0, t->blockNpixPart, t->stream);
    }
    t->blockFrames = 0;
}
void join_aux(mipt_tracer* t) {
    flush_block(t);
    if (t->pipeBusy && t->fold && t->joinEv) {
        // a pipelined frame ends with its fold on `fold` (in frame order), or -- summed straight into the caller's
        // buffer -- with its integrator on `stream` / `aux`
        hipEventRecord(t->joinEv, t->fold);
        hipStreamWaitEvent(t->stream, t->joinEv, 0);
        if (t->joinEv2) { hipEventRecord(t->joinEv2, t->aux); hipStreamWaitEvent(t->stream, t->joinEv2, 0); }
        t->pipeBusy = false;
    }
    t->mainTouched = true;
}
// every entry point except the frame pipeline itself
int use(mipt_tracer* t) { int rc = use_device(t); if (rc) return rc; join_aux(t); return MIPT_OK; }

void free_queue(Queue& q) { if (q.a) hipFree(q.a); if (q.b) hipFree(q.b); if (q.c) hipFree(q.c); q = Queue{}; }

int ensure_queue(mipt_tracer* t, int which, size_t cap) {
    if (t->qcap[which] >= cap) return MIPT_OK;
    free_queue(t->q[which]); t->qcap[which] = 0;
    if (cap == 0) return MIPT_OK;
    HIPCHK(t, hipMalloc((void**)&t->q[which].a, cap * sizeof(float4)));
    HIPCHK(t, hipMalloc((void**)&t->q[which].b, cap * sizeof(float4)));
    HIPCHK(t, hipMalloc((void**)&t->q[which].c, cap * sizeof(float4)));
    t->qcap[which] = cap;
    return MIPT_OK;
}

// Timing events come from a pool owned by the tracer (the per-frame hipEventCreate / hipEventDestroy pairs of the
// first version sat in the hot loop and leaked on an early return).
hipEvent_t take_event(mipt_tracer* t) {
    if (!t->freeEvents.empty()) { hipEvent_t e = t->freeEvents.back(); t->freeEvents.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void give_event(mipt_tracer* t, hipEvent_t e) { if (e) t->freeEvents.push_back(e); }

void resolve_events(mipt_tracer* t) {
    // kernelMs: sum of the launches' own event windows (pipelined launches overlap pairwise, so the windows add up to
    // more than the time the GPU was busy); kernelSpanMs: per resolved batch, first start to last end -- the time the
    // integrator kernels actually occupied, whichever stream they ran on
    float span = 0.f;
    for (auto& p : t->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) t->kernelMs += ms;
        if (hipEventElapsedTime(&ms, t->pending.front().first, p.second) == hipSuccess && ms > span) span = ms;
    }
    t->kernelSpanMs += span;
    for (auto& p : t->pending) { give_event(t, p.first); give_event(t, p.second); }
    t->pending.clear();
    for (auto& p : t->levelPending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.level >= 0) { t->levelMs[p.level] += ms; t->levelLaunches[p.level]++; } else t->accumulateMs += ms;
        }
        give_event(t, p.a); give_event(t, p.b);
    }
    t->levelPending.clear();
}

// per-frame camera constants: RTProgram::GenRay, CUDATracerProg.cu:369-397
FrameDev make_frame(const mipt_settings& s, uint32_t frameID) {
    FrameDev F;
    memcpy(F.camPos, s.cam.pos, 12); memcpy(F.front, s.cam.front, 12); memcpy(F.right, s.cam.right, 12); memcpy(F.up, s.cam.up, 12);
    float num_x = (float)s.viewportWidth, num_y = (float)s.viewportHeight;
    float edgeLen = std::min(num_x, num_y);
    float fovScale = atanf(s.cam.fov / 2);                       // (sic) atan, CUDATracerProg.cu:389
    F.pixelSize = (float)(1.0 / edgeLen * fovScale * 2);
    float w = num_x * F.pixelSize, h = num_y * F.pixelSize;
    F.originX = -w / 2; F.originY = -h / 2;
    F.W = s.viewportWidth; F.H = s.viewportHeight; F.frameID = frameID; F.maxDepth = s.maxDepth;
    return F;
}

inline float hdot(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

// ---- BVH cache file: header, nodes (16 floats each), refTri (uint32 each) --------------------------------------
struct BvhCacheHeader {
    char magic[8];              // "MIPTBVH1"
    uint64_t key;               // hash of the triangle positions and the build parameters
    uint32_t nTris, nNodes, nRefs, width;
};
// Bump kBvhBuilderVersion with any change to the builders, the 4-wide collapse or the quantiser: a cache file written by
// an earlier build must not be accepted (load_bvh_cache re-validates structure, not that boxes enclose their triangles).
constexpr uint64_t kBvhBuilderVersion = 5;
uint64_t bvh_cache_key(const std::vector<TriPos>& tris, const BvhParams& bp) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t w) { h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; };
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tris.data());
    const size_t n = tris.size() * 9;
    size_t i = 0;
    for (; i + 1 < n; i += 2) mix((uint64_t)w[i] | ((uint64_t)w[i + 1] << 32));
    if (i < n) mix(w[i]);
    uint32_t a, c; memcpy(&a, &bp.splitAlpha, 4); memcpy(&c, &bp.triCost, 4);
    mix(bp.builder); mix(bp.maxLeafSize); mix(a); mix(c); mix(tris.size()); mix(bp.nodeWidth);
    mix(kBvhBuilderVersion); mix(bvh_build_constants_hash());
    return h;
}
bool load_bvh_cache(const char* path, uint64_t key, uint32_t nTris, uint32_t width, BvhOut& bo) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    BvhCacheHeader h{};
    bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, "MIPTBVH1", 8) && h.key == key && h.nTris == nTris && (width ? h.width == width : (h.width == 2 || h.width == 4)) &&
              h.nNodes > 0 && h.nNodes < (1u << 27) && h.nRefs < (1u << 27);
    if (ok) {
        bo.nodes.resize((size_t)h.nNodes * 16); bo.refTri.resize(h.nRefs);
        ok = fread(bo.nodes.data(), 64, h.nNodes, f) == h.nNodes && (h.nRefs == 0 || fread(bo.refTri.data(), 4, h.nRefs, f) == h.nRefs);
    }
    fclose(f);
    bo.width = h.width;
    return ok && validate_bvh(bo, nTris).empty();
}
// Sampling tables of the env-map importance sampling estimator (option "estimator" = 1): rowCdf[H+1], then H rows of
// colCdf[W+1].  Specification (include/mipt.h; the test oracle restates it independently): weight(iv,iu) = ((double)r + g + b) *
// sin(pi (iv + 0.5) / H); per row the running sum over the row total (uniform if the row is black); the marginal
// likewise over the row totals; all in double, stored as float.
void build_env_cdf(const float* env, uint32_t W, uint32_t H, std::vector<float>& out) {
    out.assign((size_t)(H + 1) + (size_t)H * (W + 1), 0.f);
    float* row = out.data(); float* col = row + (H + 1);
    std::vector<double> rowTotal(H), acc(W + 1);
    for (uint32_t iv = 0; iv < H; iv++) {
        const double sinT = sin(M_PI * ((double)iv + 0.5) / (double)H);
        acc[0] = 0.0;
        for (uint32_t iu = 0; iu < W; iu++) {
            const float* p = env + ((size_t)iv * W + iu) * 3;
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
}

// Emitter table of estimator 1 (include/mipt.h): every triangle, in global id order, of a mesh with emissiveStrength > 0,
// a non-black emissive colour and a non-zero area.  In double from the float vertices: c = cross(e1, e2),
// area = |c| / 2, unit normal = c / |c|; CDF = running area / total area as float; Le = colour * strength in float.
float build_emitters(const mipt_scene_desc* d, const std::vector<TriPos>& tris, const std::vector<uint32_t>& triMesh,
                     std::vector<float>& tab, std::vector<float>& cdf) {
    tab.clear(); cdf.clear();
    std::vector<double> area;
    for (size_t g = 0; g < tris.size(); g++) {
        const mipt_material& mt = d->meshes[triMesh[g]].material;
        if (!(mt.emissiveStrength > 0.f) || !(mt.emissiveColor[0] > 0.f || mt.emissiveColor[1] > 0.f || mt.emissiveColor[2] > 0.f)) continue;
        const TriPos& tp = tris[g];
        float e1[3], e2[3];
        for (int a = 0; a < 3; a++) { e1[a] = tp.v[1][a] - tp.v[0][a]; e2[a] = tp.v[2][a] - tp.v[0][a]; }
        const double cx = (double)e1[1] * e2[2] - (double)e2[1] * e1[2], cy = (double)e1[2] * e2[0] - (double)e2[2] * e1[0],
                     cz = (double)e1[0] * e2[1] - (double)e2[0] * e1[1];
        const double len = sqrt(cx * cx + cy * cy + cz * cz);
        if (!(len > 0.0)) continue;
        const float rec[15] = { tp.v[0][0], tp.v[0][1], tp.v[0][2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2],
                                (float)(cx / len), (float)(cy / len), (float)(cz / len),
                                mt.emissiveColor[0] * mt.emissiveStrength, mt.emissiveColor[1] * mt.emissiveStrength, mt.emissiveColor[2] * mt.emissiveStrength };
        tab.insert(tab.end(), rec, rec + 15);
        area.push_back(0.5 * len);
    }
    if (area.empty()) return 0.f;
    double total = 0.0; for (double a : area) total += a;
    cdf.resize(area.size() + 1);
    double run = 0.0;
    for (size_t i = 0; i <= area.size(); i++) { cdf[i] = (float)(run / total); if (i < area.size()) run += area[i]; }
    cdf[area.size()] = 1.0f;
    return (float)(total / (2.0 * M_PI * 0.98));
}

bool save_bvh_cache(const char* path, uint64_t key, uint32_t nTris, const BvhOut& bo) {
    std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    BvhCacheHeader h; memcpy(h.magic, "MIPTBVH1", 8); h.key = key; h.nTris = nTris;
    h.nNodes = (uint32_t)(bo.nodes.size() / 16); h.nRefs = (uint32_t)bo.refTri.size(); h.width = bo.width;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(bo.nodes.data(), 64, h.nNodes, f) == h.nNodes &&
              (h.nRefs == 0 || fwrite(bo.refTri.data(), 4, h.nRefs, f) == h.nRefs);
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path) == 0;
    if (!ok) remove(tmp.c_str());
    return ok;
}

// flattens the meshes of a desc: triangle positions, mesh id per triangle, and the 12-float shading record per triangle
// chunks [b, e) of [0, n) on up to 16 host threads ($MIPT_BUILD_THREADS); small ranges run on the caller's thread
template <class F> void parallel_chunks(size_t n, F f) {
    int threads = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MIPT_BUILD_THREADS")) threads = std::atoi(e);
    threads = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if (n < (1u << 16) || threads == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t step = (n + threads - 1) / threads;
    try {
        for (size_t b = step; b < n; b += step) th.emplace_back(f, b, std::min(n, b + step));
    } catch (...) { for (auto& x : th) x.join(); throw; }
    f((size_t)0, std::min(n, step));
    for (auto& x : th) x.join();
}

int gather_geometry(mipt_tracer* t, const mipt_scene_desc* d, std::vector<TriPos>& tris, std::vector<uint32_t>& triMesh, std::vector<float>& shade) {
    if (d->meshCnt && !d->meshes) return fail(t, MIPT_E_INVALID, "meshes is null");
    size_t total = 0;
    for (uint32_t m = 0; m < d->meshCnt; m++) {
        const mipt_mesh_desc& md = d->meshes[m];
        if (md.indicesCnt % 3) return fail(t, MIPT_E_INVALID, "indicesCnt is not a multiple of 3");
        if (md.indicesCnt && (!md.indices || !md.vertices)) return fail(t, MIPT_E_INVALID, "mesh arrays are null");
        total += md.indicesCnt / 3;
    }
    tris.resize(total); triMesh.resize(total); shade.resize(total * 12);
    size_t base = 0;
    for (uint32_t m = 0; m < d->meshCnt; m++) {
        const mipt_mesh_desc& md = d->meshes[m];
        const size_t nt = md.indicesCnt / 3;
        std::atomic<int> bad{0};          // 1: vertex index out of range, 2: coordinate not finite or beyond 2^40
        parallel_chunks(nt, [&, m, base](size_t b, size_t e) {
            for (size_t k3 = b; k3 < e; k3++) {
                TriPos tp; float nn[3][3];
                for (int k = 0; k < 3; k++) {
                    const uint32_t vi = md.indices[k3 * 3 + k];
                    if (vi >= md.vertexCnt) { bad.store(1); return; }
                    memcpy(tp.v[k], md.vertices[vi].position, 12);
                    // finite and below 2^40: the integrator's short reciprocals (kernels.hip rcp_newton) are bit-exact for
                    // |det| <= 2^100, which unit-scale directions and edges of at most 2^41 keep far away
                    for (int a = 0; a < 3; a++) if (!(std::fabs(tp.v[k][a]) < 1.099511627776e12f)) { bad.store(2); return; }
                    // unit vertex normal, as the reference normalises at hit time (CUDATracerProg.cu:290-293)
                    const float* n = md.vertices[vi].normal;
                    float len = sqrtf(hdot(n, n));
                    for (int a = 0; a < 3; a++) nn[k][a] = (n[a] * 1.f) / len;
                }
                const size_t g = base + k3;
                tris[g] = tp; triMesh[g] = m;
                float rec[12] = { nn[0][0], nn[0][1], nn[0][2], 0, nn[1][0], nn[1][1], nn[1][2], 0, nn[2][0], nn[2][1], nn[2][2], 0 };
                memcpy(&rec[3], &m, 4);
                memcpy(&shade[g * 12], rec, 48);
            }
        });
        if (bad.load() == 1) return fail(t, MIPT_E_INVALID, "vertex index out of range");
        if (bad.load() == 2) return fail(t, MIPT_E_INVALID, "vertex coordinate not finite or beyond 2^40");
        base += nt;
    }
    return MIPT_OK;
}

// reference records (base vertex, two edges, triangle id, mesh id) in leaf order, and the 12-float material records
// a mesh whose hits can add emission (CUDATracerProg.cu:528 adds emissiveColor * emissiveStrength at every hit)
bool mesh_emits(const mipt_material& m) {
    return m.emissiveStrength != 0.f && (m.emissiveColor[0] != 0.f || m.emissiveColor[1] != 0.f || m.emissiveColor[2] != 0.f);
}
uint32_t count_emissive_meshes(const mipt_scene_desc* d) {
    uint32_t n = 0;
    for (uint32_t m = 0; m < d->meshCnt; m++) if (mesh_emits(d->meshes[m].material)) n++;
    return n;
}
// The last word of a reference record (e2.w) is a flag word: bit 1 = the triangle's mesh emits.  The integrator reads it
// with the triangle it has fetched anyway, so a ray that ends on a non-emitting surface at the last level needs no
// shading-record / material fetch to know that it adds nothing.
void fill_refs(const mipt_scene_desc* d, const std::vector<uint32_t>& refTri, const std::vector<TriPos>& tris, const std::vector<uint32_t>& triMesh, std::vector<float>& refs) {
    refs.resize(refTri.size() * 12);
    parallel_chunks(refTri.size(), [&](size_t b, size_t e) {
        for (size_t r = b; r < e; r++) {
            uint32_t g = refTri[r]; const TriPos& tp = tris[g];
            float* R = &refs[r * 12];
            for (int a = 0; a < 3; a++) { R[a] = tp.v[0][a]; R[4 + a] = tp.v[1][a] - tp.v[0][a]; R[8 + a] = tp.v[2][a] - tp.v[0][a]; }
            const uint32_t flags = mesh_emits(d->meshes[triMesh[g]].material) ? 2u : 0u;
            memcpy(&R[3], &g, 4); memcpy(&R[7], &triMesh[g], 4); memcpy(&R[11], &flags, 4);
        }
    });
}
// ---- the emitters' tree ("emitters first", kernels.h SceneDev::emitRoot) ---------------------------------------------------------
// Built over the triangles of emitting meshes: up to 15 of them are one leaf (no node at all), more get a tree of their own from
// the host builder in the main tree's node layout.  Returns false -- no emitters' tree, the kernels trace last-level rays as before --
// when nothing or everything emits, or when the tree would not fit what the main tree fits (traversal stack, LDS window).
struct EmitTree { BvhOut bvh; std::vector<uint32_t> ids; uint32_t stackNeed = 0; };
// Whether a scene with `nEmit` emitting triangles (of nTris) gets the second tree.  Under a black sky always: a last-level ray
// that reaches no emitter is finished without touching the scene.  Under a lit sky the ray still has to be walked through the
// scene (hit-or-miss), so the gain is what "first hit on a non-emitter ends it" saves over the closest hit, against one more leaf
// to test per ray: measured 1.29x on a 34 k-triangle soup with 600 emitting triangles and 0.96x on the Default scene with a
// one-quad lamp (profiles/r3/exp_emitters_first.log) -- so a lit scene gets it only when the emitters are a tree of their own.
// $MIPT_EMITTER_TREE = 0 / 1 overrides (never / whenever it fits).
bool emit_tree_wanted(const mipt_scene_desc* d, size_t nEmit, size_t nTris) {
    if (nEmit == 0 || nEmit == nTris) return false;
    if (const char* e = std::getenv("MIPT_EMITTER_TREE")) { if (e[0] == '0') return false; if (e[0] == '1') return true; }
    const bool skyBlack = !(d->envMap && d->envW && d->envH) && d->sky.intensity == 0.f;
    return skyBlack || nEmit > 15;
}
bool make_emit_tree(const mipt_scene_desc* d, const std::vector<TriPos>& tris, const std::vector<uint32_t>& triMesh, const BvhParams& bpMain,
                    uint32_t mainWidth, uint32_t mainNodes, uint32_t mainRefs, uint32_t mainStackNeed, EmitTree& et) {
    et = EmitTree();
#ifdef MIPT_EXP_PAIR
    return false;       // (the pair-layout experiment keeps one tree)
#endif
    for (uint32_t g = 0; g < (uint32_t)tris.size(); g++) if (mesh_emits(d->meshes[triMesh[g]].material)) et.ids.push_back(g);
    if (!emit_tree_wanted(d, et.ids.size(), tris.size())) return false;
    if (et.ids.size() <= 15) {
        et.bvh.refTri.resize(et.ids.size());
        for (uint32_t i = 0; i < (uint32_t)et.ids.size(); i++) et.bvh.refTri[i] = i;
        et.bvh.width = mainWidth;
        et.stackNeed = 2;                                    // the sentinel and the main root
    } else {
        std::vector<TriPos> sub(et.ids.size());
        for (size_t i = 0; i < et.ids.size(); i++) sub[i] = tris[et.ids[i]];
        BvhParams bp = bpMain;
        bp.builder = bpMain.builder == MIPT_BVH_MEDIAN ? MIPT_BVH_MEDIAN : MIPT_BVH_SBVH;     // (host builders; the set is small)
        bp.nodeWidth = mainWidth;
        if (!build_bvh(sub, bp, et.bvh).empty() || et.bvh.width != mainWidth) return false;
        et.stackNeed = et.bvh.stackNeed + 1;                 // + the main root waiting below
    }
    const bool wide = mainWidth == 4;
    if (et.stackNeed > (wide ? kMaxWideStack : kMaxBvhDepth)) return false;
    // a main tree that is traversed from LDS stays there: both trees must fit the window and the 8-entry LDS stack
    const bool mainInLds = !wide && mainNodes * 4 + mainRefs * 3 <= kLdsSceneFloat4 && mainStackNeed <= 8;
    if (mainInLds) {
        const size_t all = (size_t)(mainNodes + et.bvh.nodes.size() / 16) * 4 + (size_t)(mainRefs + et.bvh.refTri.size()) * 3;
        if (all > kLdsSceneFloat4 || et.stackNeed > 8) return false;
    }
    if ((uint64_t)mainRefs + et.bvh.refTri.size() >= (1u << 27)) return false;
    return true;
}
// the emitters' tree as it lies behind the main tree: node indices and leaf ranges moved up, references as global triangle ids
uint32_t place_emit_tree(const EmitTree& et, const mipt_scene_desc* d, const std::vector<TriPos>& tris, const std::vector<uint32_t>& triMesh,
                         uint32_t mainNodes, uint32_t mainRefs, std::vector<float>& nodes, std::vector<float>& refs) {
    nodes = et.bvh.nodes;
    const int fan = et.bvh.width == 4 ? 4 : 2;
    for (size_t n = 0; n < nodes.size() / 16; n++) {
        uint32_t r[4]; memcpy(r, &nodes[n * 16 + 12], 16);
        for (int c = 0; c < fan; c++) {
            if (!(r[c] & kLeafFlag)) r[c] += mainNodes;
            else if ((r[c] >> 27) & 0xFu) r[c] = make_leaf_ref((r[c] & 0x07FFFFFFu) + mainRefs, (r[c] >> 27) & 0xFu);     // (empty slots stay kEmptyRef)
        }
        memcpy(&nodes[n * 16 + 12], r, 16);
    }
    std::vector<uint32_t> refTri(et.bvh.refTri.size());
    for (size_t i = 0; i < refTri.size(); i++) refTri[i] = et.ids[et.bvh.refTri[i]];
    fill_refs(d, refTri, tris, triMesh, refs);
    return nodes.empty() ? make_leaf_ref(mainRefs, (uint32_t)refTri.size()) : mainNodes;
}

void fill_materials(const mipt_scene_desc* d, std::vector<float>& mats) {
    mats.assign((size_t)std::max<uint32_t>(d->meshCnt, 1) * 12, 0.f);
    for (uint32_t m = 0; m < d->meshCnt; m++) {
        const mipt_material& mm = d->meshes[m].material; float* M = &mats[(size_t)m * 12];
        M[0] = mm.color[0]; M[1] = mm.color[1]; M[2] = mm.color[2]; M[3] = mm.roughness;
        M[4] = mm.metallic; M[5] = mm.opacity; M[6] = mm.n; M[7] = mm.emissiveStrength;
        M[8] = mm.emissiveColor[0]; M[9] = mm.emissiveColor[1]; M[10] = mm.emissiveColor[2];
        // Schlick's r0 (CUDATracerProg.cu:463-481), the same fp32 operations the kernel used to run per hit: it depends on the
        // material alone -- entering or leaving only swaps n1 and n2, which negates the quotient exactly before it is squared
        float n2 = mm.n; n2 += FLT_EPSILON;
        float r0 = (1.f - n2) / (1.f + n2); r0 *= r0;
        M[11] = r0;
    }
}

// Pair layout of the reference records for k_wavestack (kernels.h SceneDev::nodes2 / refs2): every leaf's references are
// rewritten as ceil(count / 2) records of 24 floats holding two triangles component by component, an odd leaf's last
// record is completed by a zero-area pad (e1 = e2 = 0: |det| = 0 < eps, never accepted), and a copy of the node array
// carries the pair index in the low 27 bits of each leaf reference.  Same triangles, same order within a leaf.
void build_pairs(const std::vector<float>& nodes, const std::vector<float>& refs, uint32_t width, std::vector<float>& nodes2, std::vector<float>& refs2) {
    nodes2 = nodes;
    refs2.clear();
    refs2.reserve(refs.size() + 24);
    const size_t nNodes = nodes.size() / 16;
    const uint32_t nChild = width == 4 ? 4u : 2u;
    for (size_t n = 0; n < nNodes; n++) {
        for (uint32_t c = 0; c < nChild; c++) {
            uint32_t ref; memcpy(&ref, &nodes[n * 16 + 12 + c], 4);
            if (!(ref & 0x80000000u)) continue;
            const uint32_t cnt = (ref >> 27) & 0xFu, first = ref & 0x07FFFFFFu;
            if (!cnt) continue;
            const uint32_t pair0 = (uint32_t)(refs2.size() / 24);
            for (uint32_t k = 0; k < cnt; k += 2) {
                float rec[24] = {};
                const uint32_t padId = 0x7FFFFFFFu;
                for (uint32_t h = 0; h < 2; h++) {
                    if (k + h < cnt) {
                        const float* R = &refs[(size_t)(first + k + h) * 12];
                        rec[0 + h] = R[0]; rec[2 + h] = R[1]; rec[4 + h] = R[2]; rec[6 + h] = R[3];            // v0, triangle id
                        rec[8 + h] = R[4]; rec[10 + h] = R[5]; rec[12 + h] = R[6]; rec[14 + h] = R[11];        // e1, flags
                        rec[16 + h] = R[8]; rec[18 + h] = R[9]; rec[20 + h] = R[10];                            // e2
                    } else memcpy(&rec[6 + h], &padId, 4);
                }
                refs2.insert(refs2.end(), rec, rec + 24);
            }
            const uint32_t ref2 = 0x80000000u | (cnt << 27) | pair0;
            memcpy(&nodes2[n * 16 + 12 + c], &ref2, 4);
        }
    }
    if (refs2.empty()) refs2.resize(24, 0.f);
}

// include/mipt.h promises that no entry point throws: every body that can allocate on the host (vectors, strings,
// the builder's worker threads) runs inside this guard.
template <typename F> int guarded(mipt_tracer* t, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { if (t) t->err = "out of host memory"; } catch (...) {} return MIPT_E_NOMEM; }
    catch (const std::exception& e) { try { if (t) t->err = std::string("exception: ") + e.what(); } catch (...) {} return MIPT_E_INVALID; }
    catch (...) { try { if (t) t->err = "unknown exception"; } catch (...) {} return MIPT_E_INVALID; }
}

}  // namespace

extern "C" {

uint32_t mipt_abi_version(void) { return MIPT_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------------------------
int mipt_create(int device, mipt_tracer** out) {
    return guarded(nullptr, [&]() -> int {
        if (!out) return MIPT_E_INVALID;
        *out = nullptr;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MIPT_E_NODEVICE;
        if (device < 0 || device >= n) return MIPT_E_INVALID;
        mipt_tracer* t = new mipt_tracer();
        t->device = device;
        if (hipSetDevice(device) != hipSuccess) { delete t; return MIPT_E_HIP; }
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) t->numCU = cu;
        if (hipStreamCreateWithFlags(&t->own, hipStreamNonBlocking) != hipSuccess) { delete t; return MIPT_E_HIP; }
        t->stream = t->own;
        // (all three streams at normal priority: a high-priority fold stream bought nothing, and when two processes share
        // one GPU its queue, parked on an event wait, kept the queue that had to signal it from being scheduled)
        bool ok = hipStreamCreateWithFlags(&t->aux, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&t->fold, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&t->joinEv, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&t->joinEv2, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&t->blockEvA, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&t->blockEvB, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&t->blockEvF, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < mipt_tracer::kRing && ok; i++)
            ok = hipEventCreateWithFlags(&t->kernEv[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&t->foldEv[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            mipt_destroy(t); return MIPT_E_HIP;
        }
        if (hipMalloc((void**)&t->counts, 64 * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&t->stats, 40 * sizeof(unsigned long long)) != hipSuccess) {
            mipt_destroy(t); return MIPT_E_NOMEM;
        }
        // this device's copy of `scratch` (each GPU holds its own, at its own address); `device` is current here
        void* sp = nullptr;
        if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) { mipt_destroy(t); return MIPT_E_HIP; }
        t->staticScratch = sp;
        hipMemsetAsync(t->stats, 0, 40 * sizeof(unsigned long long), t->stream);
        // A stream gets its hardware queue at its first command, and a process has few of them (GPU_MAX_HW_QUEUES, 4 by
        // default): streams beyond that share queues, and two frames on one queue cannot overlap.  Take the queues of the
        // pipeline's streams now, before whatever the host application creates later (an RCCL communicator created first
        // cost the frame pipeline its overlap: 6.2 -> 7.1 ms per Cornell frame).
        hipMemsetAsync(t->counts, 0, 64 * sizeof(uint32_t), t->aux);
        hipEventRecord(t->joinEv, t->fold);
        hipStreamSynchronize(t->stream); hipStreamSynchronize(t->aux); hipStreamSynchronize(t->fold);
        *out = t;
        return MIPT_OK;
    });
}

void mipt_destroy(mipt_tracer* t) {
    if (!t) return;
    hipSetDevice(t->device);
    join_aux(t);                     // an open block of direct sum-mode frames still reaches the caller's buffer
    if (t->stream) hipStreamSynchronize(t->stream);
    if (t->aux) hipStreamSynchronize(t->aux);
    if (t->fold) hipStreamSynchronize(t->fold);
    resolve_events(t);
    free_queue(t->q[0]); free_queue(t->q[1]); free_queue(t->wstack); free_queue(t->wstack2);
    for (int i = 1; i < mipt_tracer::kRing; i++) if (t->ring[i]) hipFree(t->ring[i]);
    if (t->counts2) hipFree(t->counts2);
    if (t->ringCounts) hipFree(t->ringCounts);
    for (int i = 0; i < mipt_tracer::kRing; i++) { if (t->kernEv[i]) hipEventDestroy(t->kernEv[i]); if (t->foldEv[i]) hipEventDestroy(t->foldEv[i]); }
    if (t->joinEv) hipEventDestroy(t->joinEv);
    if (t->joinEv2) hipEventDestroy(t->joinEv2);
    for (hipEvent_t e : { t->blockEvA, t->blockEvB, t->blockEvF }) if (e) hipEventDestroy(e);
    if (t->blockSum) hipFree(t->blockSum);
    for (hipEvent_t e : t->freeEvents) hipEventDestroy(e);
    if (t->fold) hipStreamDestroy(t->fold);
    if (t->aux) hipStreamDestroy(t->aux);
    if (t->counts) hipFree(t->counts);
    if (t->stats) hipFree(t->stats);
    if (t->sample) hipFree(t->sample);
    if (t->pixelRays) hipFree(t->pixelRays);
    if (t->own) hipStreamDestroy(t->own);
    delete t;
}

const char* mipt_last_error(const mipt_tracer* t) { return t ? t->err.c_str() : "null tracer"; }

int mipt_set_stream(mipt_tracer* t, void* s) {
    return guarded(t, [&]() -> int {
        if (!t) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;                   // the old stream is ordered after the pipeline's aux stream first
        hipStream_t ns = s ? (hipStream_t)s : t->own;
        if (ns != t->stream) hipStreamSynchronize(t->stream);
        t->stream = ns;
        return MIPT_OK;
    });
}

int mipt_synchronize(mipt_tracer* t) {
    return guarded(t, [&]() -> int {
        if (!t) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        HIPCHK(t, hipStreamSynchronize(t->stream));
        resolve_events(t);
        unsigned long long flag = 0;
        HIPCHK(t, hipMemcpy(&flag, t->stats + 39, sizeof flag, hipMemcpyDeviceToHost));
        if (flag) {
            hipMemsetAsync(t->stats + 39, 0, sizeof flag, t->stream);
            hipStreamSynchronize(t->stream);
            return fail(t, MIPT_E_LIMIT, "wave stack overflow: a frame lost rays (raise the per-wave capacity)");
        }
        return MIPT_OK;
    });
}

int mipt_set_option(mipt_tracer* t, const char* key, int64_t v) {
    return guarded(t, [&]() -> int {
        if (!t || !key) return MIPT_E_INVALID;
        std::string k(key);
        if (k == "tile_pixels") { if (v < 0) return fail(t, MIPT_E_INVALID, "tile_pixels < 0"); t->tilePixels = v; }
        else if (k == "queue_budget_mb") { if (v < 1) return fail(t, MIPT_E_INVALID, "queue_budget_mb < 1"); t->queueBudgetMB = v; }
        else if (k == "rng_mode") { if (v != MIPT_RNG_NODE && v != MIPT_RNG_DFS) return fail(t, MIPT_E_INVALID, "bad rng_mode"); t->rngMode = v; }
        else if (k == "count_pixel_rays") t->countPixelRays = v ? 1 : 0;
        else if (k == "refill_lanes") { if (v < 1 || v > 64) return fail(t, MIPT_E_INVALID, "refill_lanes out of range"); t->refillLanes = v; }
        else if (k == "min_descend_lanes") { if (v < 0 || v > 64) return fail(t, MIPT_E_INVALID, "min_descend_lanes out of range"); t->minDescend = v; }
    else if (k == "xcd_bands") { if (v < -1 || v > 1) return fail(t, MIPT_E_INVALID, "xcd_bands must be -1 (auto), 0 or 1"); t->xcdBands = v; }
        else if (k == "pipeline_frames") t->pipelineFrames = v ? 1 : 0;
        else if (k == "keep_sample") t->keepSample = v ? 1 : 0;
        else if (k == "wave_stack_entries") { if (v < 0 || v > (1 << 20)) return fail(t, MIPT_E_INVALID, "wave_stack_entries out of range"); t->waveStackEntries = v; }
        else if (k == "estimator") { if (v < 0 || v > 3) return fail(t, MIPT_E_INVALID, "estimator must be 0 (reference), 1 (light sampling), 2 (VNDF + cosine lobes) or 3 (both)"); t->estimator = v; }
        else if (k == "profile_levels") t->profileLevels = v ? 1 : 0;
        else if (k == "integrator") { if (v != 0 && v != 1) return fail(t, MIPT_E_INVALID, "integrator must be 0 (wave stacks) or 1 (level queues)"); t->integrator = v; }
        else if (k == "blocks_per_cu") { if (v < 1 || v > 32) return fail(t, MIPT_E_INVALID, "blocks_per_cu out of range"); t->blocksPerCU = v; }
        else return fail(t, MIPT_E_INVALID, "unknown option " + k);
        return MIPT_OK;
    });
}

int mipt_set_tile_partition(mipt_tracer* t, uint32_t tile_rows, uint32_t n_parts, uint32_t part) {
    return guarded(t, [&]() -> int {
        if (!t) return MIPT_E_INVALID;
        if (n_parts == 0 || part >= n_parts || (n_parts > 1 && tile_rows == 0)) return fail(t, MIPT_E_INVALID, "tile partition: need tile_rows >= 1 and part < n_parts");
        t->tileRows = tile_rows; t->tileParts = n_parts; t->tilePart = part;
        return MIPT_OK;
    });
}

int mipt_get_counters(mipt_tracer* t, mipt_counters* out) {
    return guarded(t, [&]() -> int {
        if (!t || !out) return MIPT_E_INVALID;
        int rc = mipt_synchronize(t); if (rc) return rc;
        unsigned long long rays = 0;
        HIPCHK(t, hipMemcpy(&rays, t->stats, sizeof rays, hipMemcpyDeviceToHost));
        unsigned long long shadow = 0;
        HIPCHK(t, hipMemcpy(&shadow, t->stats + 38, sizeof shadow, hipMemcpyDeviceToHost));
        out->rays = rays; out->frames = t->frames; out->kernel_ms = t->kernelMs; out->launches = t->launches; out->shadow_rays = shadow; out->kernel_span_ms = t->kernelSpanMs;
        return MIPT_OK;
    });
}

int mipt_reset_counters(mipt_tracer* t) {
    return guarded(t, [&]() -> int {
        if (!t) return MIPT_E_INVALID;
        int rc = mipt_synchronize(t); if (rc) return rc;
        HIPCHK(t, hipMemsetAsync(t->stats, 0, 40 * sizeof(unsigned long long), t->stream));
        HIPCHK(t, hipStreamSynchronize(t->stream));
        t->frames = 0; t->launches = 0; t->kernelMs = 0; t->kernelSpanMs = 0;
        for (int i = 0; i < 32; i++) { t->levelMs[i] = 0; t->levelLaunches[i] = 0; }
        t->accumulateMs = 0;
        return MIPT_OK;
    });
}

int mipt_get_level_profile(mipt_tracer* t, mipt_level_profile* out) {
    return guarded(t, [&]() -> int {
        if (!t || !out) return MIPT_E_INVALID;
        int rc = mipt_synchronize(t); if (rc) return rc;
        unsigned long long st[40];
        HIPCHK(t, hipMemcpy(st, t->stats, sizeof st, hipMemcpyDeviceToHost));
        for (int i = 0; i < 32; i++) { out->rays[i] = st[1 + i]; out->ms[i] = t->levelMs[i]; out->launches[i] = t->levelLaunches[i]; }
        out->accumulate_ms = t->accumulateMs;
        return MIPT_OK;
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Scene upload.  One flat, aligned buffer per stream (nodes / refs / shading / materials / env map) instead of the
// reference's per-node cudaMemcpy walk (CUDATracer.hpp:99-129).
// ---------------------------------------------------------------------------------------------------------------
int mipt_scene_create(mipt_tracer* t, const mipt_scene_desc* d, mipt_scene** out) {
    return guarded(t, [&]() -> int {
        if (!d || !out) return MIPT_E_INVALID;
        *out = nullptr;
        if (t && use(t)) return MIPT_E_HIP;     // t == NULL: host-only scene (BVH build + mipt_scene_get_bvh, no upload)
        std::vector<TriPos> tris;
        std::vector<uint32_t> triMesh;
        std::vector<float> shade;     // 12 per triangle
        const bool verbose = std::getenv("MIPT_BUILD_VERBOSE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tStart = now();
        if (int rc = gather_geometry(t, d, tris, triMesh, shade)) return rc;
        const double tGather = now();
        BvhParams bp;
        bp.builder = d->bvhBuilder;
        if (bp.builder != MIPT_BVH_SBVH && bp.builder != MIPT_BVH_MEDIAN && bp.builder != MIPT_BVH_LBVH_GPU) return fail(t, MIPT_E_INVALID, "unknown bvhBuilder");
        if (bp.builder == MIPT_BVH_LBVH_GPU && !t) return fail(t, MIPT_E_INVALID, "bvhBuilder MIPT_BVH_LBVH_GPU needs a tracer (a GPU to build on)");
        if (d->maxLeafSize) bp.maxLeafSize = d->maxLeafSize;
        if (d->splitAlpha > 0.f) bp.splitAlpha = d->splitAlpha;
        // SAH triangle cost: measured optima are 1.5 for a scene traversed as BVH2 from LDS (a few dozen triangles) and 3.0
        // for everything that ends up in 4-wide nodes (profiles/r1/exp_sah_tricost_leafsize.log, exp_sah_q4.log)
        bp.triCost = d->sahTriCost > 0.f ? d->sahTriCost : (tris.size() <= 64 ? 1.5f : 3.0f);
        bp.nodeWidth = d->nodeWidth;
        if (bp.nodeWidth != 0 && bp.nodeWidth != 2 && bp.nodeWidth != 4) return fail(t, MIPT_E_INVALID, "nodeWidth must be 0, 2 or 4");
        BvhOut bo;
        const bool useCache = d->bvhCachePath && d->bvhCachePath[0];
        const uint64_t cacheKey = useCache ? bvh_cache_key(tris, bp) : 0;
        const bool cached = useCache && load_bvh_cache(d->bvhCachePath, cacheKey, (uint32_t)tris.size(), bp.nodeWidth, bo);
        if (!cached) {
            bo = BvhOut();
            std::string berr;
            if (bp.builder == MIPT_BVH_LBVH_GPU) {
                // built on the device, then held to the same structural check as a cache file before any kernel walks it:
                // every index in range, one tree rooted at node 0, every triangle referenced, stack bound (sets stackNeed)
                GpuBuildTimes gt;
                berr = build_bvh_gpu(tris, bp, bo, t->stream, &gt);
                if (berr.empty()) berr = validate_bvh(bo, (uint32_t)tris.size());
                if (berr.empty() && std::getenv("MIPT_BUILD_VERBOSE"))
                    fprintf(stderr, "[mipt] GPU BVH build: %zu triangles -> %zu %s nodes; upload %.1f ms, codes + sort %.1f, tree + boxes + BVH2 %.1f, 4-wide collapse %.1f, download %.1f\n",
                            tris.size(), bo.nodes.size() / 16, bo.width == 4 ? "4-wide" : "BVH2", gt.uploadMs, gt.sortMs, gt.treeMs, gt.collapseMs, gt.downloadMs);
            } else berr = build_bvh(tris, bp, bo);
            if (!berr.empty()) return fail(t, MIPT_E_LIMIT, "BVH build: " + berr);
            if (useCache && !save_bvh_cache(d->bvhCachePath, cacheKey, (uint32_t)tris.size(), bo))
                return fail(t, MIPT_E_INVALID, std::string("cannot write BVH cache ") + d->bvhCachePath);
        }

        const double tBuild = now();
        mipt_scene* s = new mipt_scene();
        s->t = t; s->nMeshes = d->meshCnt; s->nTris = (uint32_t)tris.size(); s->bvhFromCache = cached; s->bvhWidth = bo.width;
        s->hNodes = std::move(bo.nodes);
        s->hRefTri = bo.refTri;
        fill_refs(d, s->hRefTri, tris, triMesh, s->hRefs);
        std::vector<float> mats; fill_materials(d, mats);
        if (d->envMap && d->envW && d->envH) build_env_cdf(d->envMap, d->envW, d->envH, s->hEnvCdf);
        const float emitScale = build_emitters(d, tris, triMesh, s->hEmitTri, s->hEmitCdf);
        auto upload = [&](void** dst, const void* src, size_t bytes) -> bool {
            if (bytes == 0) bytes = 16;
            if (hipMalloc(dst, bytes) != hipSuccess) return false;
            if (src && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
            return true;
        };
        const double tRecords = now();
        if (shade.empty()) shade.resize(12, 0.f);
        const uint32_t mainNodes = (uint32_t)(s->hNodes.size() / 16), mainRefs = (uint32_t)bo.refTri.size();
        s->mainStackNeed = bo.stackNeed; s->bvhParams = bp;
        std::vector<float> emitNodes;
        {
            EmitTree et;
            if (make_emit_tree(d, tris, triMesh, bp, bo.width, mainNodes, mainRefs, bo.stackNeed, et)) {
                s->emitRootRef = place_emit_tree(et, d, tris, triMesh, mainNodes, mainRefs, emitNodes, s->hEmitRefs);
                s->hEmitBvh = std::move(et.bvh); s->hEmitIds = std::move(et.ids);
                bo.stackNeed = std::max(bo.stackNeed, et.stackNeed);
            }
        }
        s->emitNodeCap = (uint32_t)(emitNodes.size() / 16); s->emitRefCap = (uint32_t)(s->hEmitRefs.size() / 12);
        s->hEmitNodes = emitNodes;
        if (s->hRefs.empty()) s->hRefs.resize(12, 0.f);
        if (t) {
            // nodes and references as the kernels see them: the main tree, then the emitters' tree
            std::vector<float> nodesAll, refsAll;
            const std::vector<float>* upNodes = &s->hNodes; const std::vector<float>* upRefs = &s->hRefs;
            if (s->emitRootRef) {
                nodesAll = s->hNodes; nodesAll.insert(nodesAll.end(), emitNodes.begin(), emitNodes.end());
                refsAll.assign(s->hRefs.begin(), s->hRefs.begin() + (size_t)mainRefs * 12); refsAll.insert(refsAll.end(), s->hEmitRefs.begin(), s->hEmitRefs.end());
                upNodes = &nodesAll; upRefs = &refsAll;
            }
            bool ok = upload(&s->dNodes, upNodes->data(), upNodes->size() * 4) && upload(&s->dRefs, upRefs->data(), upRefs->size() * 4) &&
                      upload(&s->dShade, shade.data(), shade.size() * 4) && upload(&s->dMats, mats.data(), mats.size() * 4);
            if (ok && d->envMap && d->envW && d->envH) ok = upload(&s->dEnv, d->envMap, (size_t)d->envW * d->envH * 3 * 4);
            if (ok && s->dEnv) ok = upload(&s->dEnvCdf, s->hEnvCdf.data(), s->hEnvCdf.size() * 4);
            if (ok && !s->hEmitCdf.empty()) ok = upload(&s->dEmitTri, s->hEmitTri.data(), s->hEmitTri.size() * 4) && upload(&s->dEmitCdf, s->hEmitCdf.data(), s->hEmitCdf.size() * 4);
#ifdef MIPT_EXP_PAIR
            if (ok) {
                std::vector<float> nodes2, refs2;
                build_pairs(s->hNodes, s->hRefs, s->bvhWidth, nodes2, refs2);
                ok = upload(&s->dNodes2, nodes2.data(), nodes2.size() * 4) && upload(&s->dRefs2, refs2.data(), refs2.size() * 4);
                s->nodes2Bytes = nodes2.size() * 4; s->refs2Bytes = refs2.size() * 4;
            }
#endif
            if (!ok) { mipt_scene_destroy(s); return fail(t, MIPT_E_NOMEM, "scene upload failed"); }
        }
        if (bo.refTri.empty()) s->hRefs.clear();
        s->dev.nodes = (const float4*)s->dNodes; s->dev.refs = (const float4*)s->dRefs; s->dev.shade = (const float4*)s->dShade;
        s->dev.nodes2 = (const float4*)s->dNodes2; s->dev.refs2 = (const float4*)s->dRefs2;
        s->dev.mats = (const float4*)s->dMats; s->dev.env = (const float*)s->dEnv;
        s->dev.envW = s->dEnv ? d->envW : 0; s->dev.envH = s->dEnv ? d->envH : 0;
        s->dev.envCdf = (const float*)s->dEnvCdf;
        s->dev.emitTri = (const float*)s->dEmitTri; s->dev.emitCdf = (const float*)s->dEmitCdf;
        s->dev.nEmit = s->dEmitCdf ? (uint32_t)(s->hEmitCdf.size() - 1) : 0u; s->dev.emitScale = emitScale;
        s->emitCap = s->dev.nEmit;
        s->dev.neeScale = s->dEnv ? (float)(M_PI / 0.98 / ((double)d->envW * (double)d->envH)) : 0.f;
        memcpy(s->dev.skyColor, d->sky.color, 12); memcpy(s->dev.skyDir, d->sky.dir, 12); s->dev.skyIntensity = d->sky.intensity;
        s->dev.nNodes = mainNodes + s->emitNodeCap; s->dev.nRefs = mainRefs + s->emitRefCap;
        s->dev.emitRoot = s->emitRootRef; s->dev.nRefsMain = mainRefs;
        s->dev.wide = bo.width == 4 ? 1u : 0u;
        s->dev.stackNeed = bo.stackNeed;
        s->dev.emissiveMeshes = count_emissive_meshes(d);
        if (t) hipDeviceSynchronize();
        if (verbose)
            fprintf(stderr, "[mipt] scene: %zu triangles, %u nodes (width %u); gather %.0f ms, BVH %s %.0f ms, records + tables %.0f ms, emitters' tree + upload %.0f ms\n",
                    tris.size(), s->dev.nNodes, bo.width, tGather - tStart, cached ? "from cache" : "build + check", tBuild - tGather, tRecords - tBuild, now() - tRecords);
        *out = s;
        return MIPT_OK;
    });
}

void mipt_scene_destroy(mipt_scene* s) {
    if (!s) return;
    if (s->t) hipSetDevice(s->t->device);
    if (s->dNodes) hipFree(s->dNodes);
    if (s->dRefs) hipFree(s->dRefs);
    if (s->dNodes2) hipFree(s->dNodes2);
    if (s->dRefs2) hipFree(s->dRefs2);
    if (s->dShade) hipFree(s->dShade);
    if (s->dMats) hipFree(s->dMats);
    if (s->dEnv) hipFree(s->dEnv);
    if (s->dEnvCdf) hipFree(s->dEnvCdf);
    if (s->dEmitTri) hipFree(s->dEmitTri);
    if (s->dEmitCdf) hipFree(s->dEmitCdf);
    delete s;
}

// Moved vertices / edited materials on an uploaded scene (the use the reference's CUDAScene::GetMeshMutable,
// CUDATracer.hpp:217-221, points at but never makes): the tree keeps its shape, boxes are re-fitted, the flat buffers are
// overwritten in place.  The sky light is taken from the desc too; the env map stays as created.
int mipt_scene_refit(mipt_scene* s, const mipt_scene_desc* d) {
    return guarded(nullptr, [&]() -> int {
        if (!s || !d) return MIPT_E_INVALID;
        mipt_tracer* t = s->t;
        if (t && use(t)) return MIPT_E_HIP;
        if (d->meshCnt != s->nMeshes) return fail(t, MIPT_E_INVALID, "refit: mesh count differs from the scene's");
        std::vector<TriPos> tris; std::vector<uint32_t> triMesh; std::vector<float> shade;
        if (int rc = gather_geometry(t, d, tris, triMesh, shade)) return rc;
        if (tris.size() != s->nTris) return fail(t, MIPT_E_INVALID, "refit: triangle count differs from the scene's");
        if (s->nTris == 0) return MIPT_OK;
        BvhOut bo; bo.nodes = s->hNodes; bo.refTri = s->hRefTri; bo.width = s->bvhWidth;
        std::string err = refit_bvh(bo, tris);
        if (!err.empty()) return fail(t, MIPT_E_LIMIT, "BVH refit: " + err);
        std::vector<float> refs, mats, emitTri, emitCdf;
        fill_refs(d, s->hRefTri, tris, triMesh, refs);
        fill_materials(d, mats);
        const float emitScale = build_emitters(d, tris, triMesh, emitTri, emitCdf);
        // the emitters' tree: the same emitting triangles -> same shape, boxes re-fitted; another set (a material edit) -> rebuilt
        // if it fits the room behind the main tree, else the scene goes on without one
        const uint32_t mainNodes = (uint32_t)(bo.nodes.size() / 16), mainRefs = (uint32_t)s->hRefTri.size();
        EmitTree et; std::vector<float> emitNodes, emitRefs; uint32_t emitRoot = 0;
        {
            std::vector<uint32_t> ids;
            for (uint32_t g = 0; g < (uint32_t)tris.size(); g++) if (mesh_emits(d->meshes[triMesh[g]].material)) ids.push_back(g);
            bool have = false;
            if (!emit_tree_wanted(d, ids.size(), tris.size())) have = false;
            else if (!s->hEmitIds.empty() && ids == s->hEmitIds) {      // (also when an edit in between had switched the tree off: its shape is kept)
                et.bvh = s->hEmitBvh; et.ids = ids; have = true;
                if (!et.bvh.nodes.empty()) {
                    std::vector<TriPos> sub(ids.size());
                    for (size_t i = 0; i < ids.size(); i++) sub[i] = tris[ids[i]];
                    have = refit_bvh(et.bvh, sub).empty();
                }
                et.stackNeed = et.bvh.nodes.empty() ? 2u : et.bvh.stackNeed + 1u;
            } else if (s->emitRefCap) {
                have = make_emit_tree(d, tris, triMesh, s->bvhParams, s->bvhWidth, mainNodes, mainRefs, s->mainStackNeed, et) &&
                       et.bvh.nodes.size() / 16 <= s->emitNodeCap && et.bvh.refTri.size() <= s->emitRefCap;
            }
            if (have) emitRoot = place_emit_tree(et, d, tris, triMesh, mainNodes, mainRefs, emitNodes, emitRefs);
        }
        if (t) {
            // every frame in flight has been joined by use(); the stream of this tracer may still be running them
            HIPCHK(t, hipStreamSynchronize(t->stream));
            HIPCHK(t, hipMemcpy(s->dNodes, bo.nodes.data(), bo.nodes.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(t, hipMemcpy(s->dRefs, refs.data(), refs.size() * 4, hipMemcpyHostToDevice));
            if (emitRoot) {
                if (!emitNodes.empty()) HIPCHK(t, hipMemcpy((char*)s->dNodes + (size_t)mainNodes * 64, emitNodes.data(), emitNodes.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(t, hipMemcpy((char*)s->dRefs + (size_t)mainRefs * 48, emitRefs.data(), emitRefs.size() * 4, hipMemcpyHostToDevice));
            }
#ifdef MIPT_EXP_PAIR
            {   // the pair layout follows (same structure, so the same sizes)
                std::vector<float> nodes2, refs2;
                build_pairs(bo.nodes, refs, s->bvhWidth, nodes2, refs2);
                if (nodes2.size() * 4 != s->nodes2Bytes || refs2.size() * 4 != s->refs2Bytes) return fail(t, MIPT_E_INVALID, "refit: pair layout changed size");
                HIPCHK(t, hipMemcpy(s->dNodes2, nodes2.data(), nodes2.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(t, hipMemcpy(s->dRefs2, refs2.data(), refs2.size() * 4, hipMemcpyHostToDevice));
            }
#endif
            HIPCHK(t, hipMemcpy(s->dShade, shade.data(), shade.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(t, hipMemcpy(s->dMats, mats.data(), mats.size() * 4, hipMemcpyHostToDevice));
            const size_t nEmit = emitCdf.empty() ? 0 : emitCdf.size() - 1;
            if (nEmit > s->emitCap) {           // a material turned emissive: the emitter table grows
                if (s->dEmitTri) HIPCHK(t, hipFree(s->dEmitTri));
                if (s->dEmitCdf) HIPCHK(t, hipFree(s->dEmitCdf));
                s->dEmitTri = s->dEmitCdf = nullptr; s->emitCap = 0;
                HIPCHK(t, hipMalloc(&s->dEmitTri, nEmit * 15 * 4));
                HIPCHK(t, hipMalloc(&s->dEmitCdf, (nEmit + 1) * 4));
                s->emitCap = nEmit;
            }
            if (nEmit) {
                HIPCHK(t, hipMemcpy(s->dEmitTri, emitTri.data(), emitTri.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(t, hipMemcpy(s->dEmitCdf, emitCdf.data(), emitCdf.size() * 4, hipMemcpyHostToDevice));
            }
            s->dev.emitTri = (const float*)s->dEmitTri; s->dev.emitCdf = (const float*)s->dEmitCdf;
            s->dev.nEmit = (uint32_t)nEmit; s->dev.emitScale = emitScale;
        }
        s->dev.emissiveMeshes = count_emissive_meshes(d);
        s->dev.emitRoot = s->emitRootRef = emitRoot;
        s->dev.stackNeed = emitRoot ? std::max(s->mainStackNeed, et.stackNeed) : s->mainStackNeed;
        if (emitRoot) { s->hEmitBvh = std::move(et.bvh); s->hEmitIds = std::move(et.ids); s->hEmitRefs.swap(emitRefs); s->hEmitNodes.swap(emitNodes); }
        // the analytic sky is a launch parameter, not a buffer: taken from the desc as well (the env map stays)
        memcpy(s->dev.skyColor, d->sky.color, 12); memcpy(s->dev.skyDir, d->sky.dir, 12); s->dev.skyIntensity = d->sky.intensity;
        s->hNodes.swap(bo.nodes); s->hRefs.swap(refs); s->hEmitTri.swap(emitTri); s->hEmitCdf.swap(emitCdf);
        return MIPT_OK;
    });
}

int mipt_scene_get_bvh(const mipt_scene* s, float* nodes, uint32_t* n_nodes, float* refs, uint32_t* n_refs) {
    return guarded(nullptr, [&]() -> int {
        if (!s) return MIPT_E_INVALID;
        if (n_nodes) *n_nodes = (uint32_t)(s->hNodes.size() / 16);
        if (n_refs) *n_refs = (uint32_t)(s->hRefs.size() / 12);
        if (nodes) memcpy(nodes, s->hNodes.data(), s->hNodes.size() * 4);
        if (refs) memcpy(refs, s->hRefs.data(), s->hRefs.size() * 4);
        return MIPT_OK;
    });
}
int mipt_scene_get_emitter_tree(const mipt_scene* s, float* nodes, uint32_t* n_nodes, float* refs, uint32_t* n_refs, uint32_t* root_ref) {
    return guarded(nullptr, [&]() -> int {
        if (!s) return MIPT_E_INVALID;
        const bool on = s->emitRootRef != 0;
        if (n_nodes) *n_nodes = on ? (uint32_t)(s->hEmitNodes.size() / 16) : 0u;
        if (n_refs) *n_refs = on ? (uint32_t)(s->hEmitRefs.size() / 12) : 0u;
        if (root_ref) *root_ref = s->emitRootRef;
        if (on && nodes && !s->hEmitNodes.empty()) memcpy(nodes, s->hEmitNodes.data(), s->hEmitNodes.size() * 4);
        if (on && refs) memcpy(refs, s->hEmitRefs.data(), s->hEmitRefs.size() * 4);
        return MIPT_OK;
    });
}
uint32_t mipt_scene_triangle_count(const mipt_scene* s) { return s ? s->nTris : 0; }
uint32_t mipt_scene_bvh_width(const mipt_scene* s) { return s ? s->bvhWidth : 0; }
uint64_t mipt_scene_get_env_cdf(const mipt_scene* s, float* out, uint64_t cap) {
    if (!s) return 0;
    if (out && cap >= s->hEnvCdf.size()) memcpy(out, s->hEnvCdf.data(), s->hEnvCdf.size() * sizeof(float));
    return s->hEnvCdf.size();
}
uint64_t mipt_scene_get_emitters(const mipt_scene* s, float* tris, float* cdf, uint64_t cap_tris) {
    if (!s) return 0;
    const uint64_t n = s->hEmitTri.size() / 15;
    if (tris && cdf && n && cap_tris >= n) { memcpy(tris, s->hEmitTri.data(), n * 15 * sizeof(float)); memcpy(cdf, s->hEmitCdf.data(), (n + 1) * sizeof(float)); }
    return n;
}
int mipt_scene_bvh_was_cached(const mipt_scene* s) { return s && s->bvhFromCache ? 1 : 0; }

// ---------------------------------------------------------------------------------------------------------------
// The hot path
// ---------------------------------------------------------------------------------------------------------------
int mipt_render(mipt_tracer* t, mipt_scene* sc, const mipt_settings* st, uint32_t first_frame, uint32_t n_frames,
                uint32_t frame_stride, int sum_mode, float* acc_dev, unsigned char* ldr_dev) {
    return guarded(t, [&]() -> int {
        if (!t || !sc || !st || !acc_dev) return fail(t, MIPT_E_INVALID, "null argument");
        if (sc->t != t) return fail(t, MIPT_E_INVALID, "scene belongs to another tracer");
        if (use_device(t)) return MIPT_E_HIP;
        const uint32_t W = st->viewportWidth, H = st->viewportHeight, D = st->maxDepth;
        if (!W || !H) return fail(t, MIPT_E_INVALID, "empty viewport");
        if ((uint64_t)W * H >= (1ull << 31)) return fail(t, MIPT_E_LIMIT, "viewport too large");
        if (D > 30) return fail(t, MIPT_E_LIMIT, "maxDepth > 30 (node ids are 32-bit)");
        if (t->rngMode == MIPT_RNG_DFS && D > 16) return fail(t, MIPT_E_LIMIT, "rng_mode dfs supports maxDepth <= 16");
        if (frame_stride == 0) frame_stride = 1;
        const uint32_t npix = W * H;
        // env-map importance sampling: wave-stack integrator with node-keyed streams only, and only with an env map to sample
        const bool nee = t->estimator == 1 || t->estimator == 3;
        if (nee && (t->rngMode != MIPT_RNG_NODE || t->integrator != 0)) return fail(t, MIPT_E_INVALID, "estimators 1 and 3 need integrator 0 and rng_mode node");
        if (nee && !sc->dev.envCdf && !sc->dev.nEmit) return fail(t, MIPT_E_INVALID, "estimators 1 and 3 need a scene with an environment map or emissive triangles");
        if (nee && npix >= (1u << 29)) return fail(t, MIPT_E_LIMIT, "viewport too large for estimator 1");
        if (t->estimator >= 2 && (t->rngMode != MIPT_RNG_NODE || t->integrator != 0)) return fail(t, MIPT_E_INVALID, "estimator 2 needs integrator 0 and rng_mode node");

        if (t->samplePixels < npix) {
            if (t->sample) hipFree(t->sample);
            t->sample = nullptr; t->samplePixels = 0;
            HIPCHK(t, hipMalloc((void**)&t->sample, (size_t)npix * 16));
            t->samplePixels = npix;
            t->slotDirty[0] = npix;
        }
        if (t->countPixelRays && t->pixelRaysCap < npix) {
            if (t->pixelRays) hipFree(t->pixelRays);
            t->pixelRays = nullptr; t->pixelRaysCap = 0;
            HIPCHK(t, hipMalloc((void**)&t->pixelRays, (size_t)npix * 4));
            t->pixelRaysCap = npix;
        }
        const int persistentBlocks = t->numCU * (int)t->blocksPerCU;
        uint32_t tile = npix;
        uint32_t wcap = 0;
        // Frame pipeline.  A wave of the persistent kernel cannot give work away, so a frame ends with a tail in which the
        // waves holding the heaviest pixels finish alone (measured: 10-20 % of a 1080p frame, scripts/tail_profile.py).
        // Consecutive frames therefore alternate between two streams with their own wave stacks and pixel counter, writing
        // into a ring of sample buffers: the next frame's blocks move onto the CUs as the previous frame's blocks retire.
        // The folds into the caller's buffers go, in frame order, onto a third, high-priority stream (behind a persistent
        // kernel's resident waves a normal-priority fold starves for milliseconds), so the running mean is folded in order
        // and no integrator launch ever queues behind a fold.
        // Off for a caller-supplied stream (the caller expects everything in order there), per-pixel ray counting, and
        // the other integrators.
        // row-tile partition: this handle renders tiles part, part + parts, ... of tile_rows rows each
        Band band{ 0u, 1u, 0u };
        uint32_t npixPart = npix;
        if (t->tileParts > 1) {
            if (t->rngMode != MIPT_RNG_NODE || t->integrator != 0 || D < 1) return fail(t, MIPT_E_INVALID, "a tile partition needs integrator 0, rng_mode node and maxDepth >= 1");
            const uint32_t W = st->viewportWidth, H = st->viewportHeight;
            if ((uint64_t)t->tileRows * W >= (1ull << 31)) return fail(t, MIPT_E_LIMIT, "tile too large");
            const uint32_t nTiles = (H + t->tileRows - 1) / t->tileRows;
            uint64_t rows = 0;
            for (uint32_t k = t->tilePart; k < nTiles; k += t->tileParts) rows += std::min(t->tileRows, H - k * t->tileRows);
            band = Band{ t->tileRows * W, t->tileParts, t->tilePart };
            npixPart = (uint32_t)(rows * W);
        }
        const bool pipe = t->pipelineFrames && t->stream == t->own && !t->countPixelRays && t->rngMode == MIPT_RNG_NODE && t->integrator == 0 && D >= 1;
        if (!pipe) join_aux(t);
        if (t->rngMode == MIPT_RNG_DFS) {
            // depth-first integrator: no queues
        } else if (t->integrator == 0) {
            // persistent wave stacks: 256 entries per recursion level per wave (see k_wavestack), independent of the viewport
            // (a capacity reached in simulation, not a proven bound: a wave that runs out of room stalls its shade phase
            // instead of overflowing, see k_wavestack A1, so the figure decides speed and not correctness)
            const uint32_t wcapDefault = (nee ? 512u : 256u) * std::max<uint32_t>(D, 1u) + 256u;      // estimator 1: up to four rays per hit
            wcap = t->waveStackEntries ? (uint32_t)t->waveStackEntries : wcapDefault;
            const size_t need = (size_t)persistentBlocks * 4 * std::max(wcap, wcapDefault);
            if (t->wstackEntries < need) {
                free_queue(t->wstack); t->wstackEntries = 0;
                HIPCHK(t, hipMalloc((void**)&t->wstack.a, need * sizeof(float4)));
                HIPCHK(t, hipMalloc((void**)&t->wstack.b, need * sizeof(float4)));
                HIPCHK(t, hipMalloc((void**)&t->wstack.c, need * sizeof(float4)));
                t->wstackEntries = need;
            }
            if (pipe) {
                if (t->wstack2Entries < need) {
                    free_queue(t->wstack2); t->wstack2Entries = 0;
                    HIPCHK(t, hipMalloc((void**)&t->wstack2.a, need * sizeof(float4)));
                    HIPCHK(t, hipMalloc((void**)&t->wstack2.b, need * sizeof(float4)));
                    HIPCHK(t, hipMalloc((void**)&t->wstack2.c, need * sizeof(float4)));
                    t->wstack2Entries = need;
                }
                if (t->ringPixels < npix) {
                    for (int i = 1; i < mipt_tracer::kRing; i++) {
                        if (t->ring[i]) hipFree(t->ring[i]);
                        t->ring[i] = nullptr;
                    }
                    t->ringPixels = 0;
                    for (int i = 1; i < mipt_tracer::kRing; i++) { HIPCHK(t, hipMalloc((void**)&t->ring[i], (size_t)npix * 16)); t->slotDirty[i] = npix; }
                    t->ringPixels = npix;
                }
                if (!t->counts2) HIPCHK(t, hipMalloc((void**)&t->counts2, 64 * sizeof(uint32_t)));
                if (!t->ringCounts) HIPCHK(t, hipMalloc((void**)&t->ringCounts, mipt_tracer::kRing * 64 * sizeof(uint32_t)));
            }
        } else if (D >= 2) {
        // tile size: rays at level k <= tile * 2^k; the two ping-pong queues hold the largest odd / even level
            uint64_t maxEntries = (uint64_t)t->queueBudgetMB * 1024 * 1024 / 48;
            uint64_t perPixel = (1ull << (D - 1)) + (D >= 3 ? (1ull << (D - 2)) : 0);
            uint64_t fit = maxEntries / perPixel;
            if (t->tilePixels > 0) fit = (uint64_t)t->tilePixels;
            if (fit < 64) return fail(t, MIPT_E_LIMIT, "queue budget too small for this maxDepth");
            if (fit < tile) tile = (uint32_t)fit;
            if (tile >= 256) tile &= ~255u;
            if (((uint64_t)tile << (D - 1)) >= (1ull << 32)) return fail(t, MIPT_E_LIMIT, "tile * 2^(maxDepth-1) exceeds 32-bit queue indices");
            uint32_t oddMax = ((D - 1) & 1) ? (D - 1) : (D - 2);          // largest odd level <= D-1 (>= 1)
            uint32_t evenMax = ((D - 1) & 1) ? (D - 2) : (D - 1);         // largest even level <= D-1 (may be 0 -> unused)
            int rc = ensure_queue(t, 1, (size_t)tile << oddMax); if (rc) return rc;
            if (evenMax >= 2) { rc = ensure_queue(t, 0, (size_t)tile << evenMax); if (rc) return rc; }
        }

        if (t->waveStackEntries && t->integrator == 0 && t->rngMode == MIPT_RNG_NODE) {
            // test option: every byte of the wave-stack arrays is painted before the call, and mipt_test_stack_guard then
            // checks that nothing beyond the waves' (shrunken) regions was written
            join_aux(t);
            HIPCHK(t, hipStreamSynchronize(t->stream));
            for (Queue* q : { &t->wstack, &t->wstack2 }) {
                const size_t n = q == &t->wstack ? t->wstackEntries : t->wstack2Entries;
                if (!n || !q->a) continue;
                // (on the tracer's own stream and waited for: a plain hipMemset goes to the null stream, which the tracer's
                // non-blocking streams do not wait for, and may still be painting when the first frame pushes its rays)
                HIPCHK(t, hipMemsetAsync(q->a, 0xA5, n * sizeof(float4), t->stream)); HIPCHK(t, hipMemsetAsync(q->b, 0xA5, n * sizeof(float4), t->stream));
                HIPCHK(t, hipMemsetAsync(q->c, 0xA5, n * sizeof(float4), t->stream));
            }
            HIPCHK(t, hipStreamSynchronize(t->stream));
            t->guardFrom = (size_t)persistentBlocks * 4 * wcap;
            t->mainTouched = true;
        }
        t->ring[0] = t->sample;
        // Per-frame side work.  The sample buffer of a frame must start at zero (the integrator adds into
        // it).  A pipelined fold now clears its slot right after reading it, so the 16 B/pixel clear per frame is gone; only a
        // slot that was left standing (option keep_sample, or a frame of one of the unpipelined paths whose sample the
        // caller may still read through mipt_last_sample) is cleared before its next use.  In sum mode without an LDR
        // target ("direct") the integrator adds straight into the tracer's block buffer, which one fold per kBlockFrames
        // frames adds to the caller's sum: no per-frame sample buffer, no per-frame fold.
        const bool direct = sum_mode && !ldr_dev && !t->countPixelRays && t->rngMode == MIPT_RNG_NODE && t->integrator == 0 && D >= 1;
        if (direct) {
            // an open block belongs to one sum buffer, one frame size, one partition and one set of streams
            if (t->blockFrames && (t->blockAcc != acc_dev || t->blockNpixPart != npixPart || t->blockBand.pixels != band.pixels || t->blockBand.stride != band.stride ||
                                   t->blockBand.first != band.first || t->blockPipe != pipe)) join_aux(t);
            if (t->blockPixels < npix) {
                join_aux(t);
                HIPCHK(t, hipStreamSynchronize(t->stream));
                if (t->blockSum) hipFree(t->blockSum);
                t->blockSum = nullptr; t->blockPixels = 0;
                HIPCHK(t, hipMalloc((void**)&t->blockSum, (size_t)npix * 16));
                HIPCHK(t, hipMemsetAsync(t->blockSum, 0, (size_t)npix * 16, t->stream));
                t->blockPixels = npix; t->mainTouched = true;
            }
            t->blockAcc = acc_dev; t->blockNpixPart = npixPart; t->blockBand = band; t->blockPipe = pipe;
        }
        if (pipe && t->mainTouched) {       // aux / fold start after whatever else the caller queued on the tracer's stream (clears, uploads ...)
            HIPCHK(t, hipEventRecord(t->joinEv, t->stream));
            HIPCHK(t, hipStreamWaitEvent(t->aux, t->joinEv, 0));
            HIPCHK(t, hipStreamWaitEvent(t->fold, t->joinEv, 0));
            t->mainTouched = false;
        }
        // An early return out of the frame loop (a HIP call failing between a launch and its fold) would leave the host's
        // notes -- "this slot's pixel counters are zero", "this slot's sample buffer is clear" -- ahead of the device: every
        // later frame on the slot would claim no pixels and render black.  Whatever leaves the loop without reaching its
        // end resets them to "clear everything before the next use".
        struct Bookkeeping {
            mipt_tracer* t; bool ok = false;
            ~Bookkeeping() {
                if (ok) return;
                for (int i = 0; i < mipt_tracer::kRing; i++) { t->ringCountsClean[i] = false; t->slotDirty[i] = i ? t->ringPixels : t->samplePixels; }
            }
        } bookkeeping{ t };
        for (uint32_t fi = 0; fi < n_frames; fi++) {
            const uint32_t frame = first_frame + fi * frame_stride;
            const int set = pipe ? (int)(t->flip & 1u) : 0;
            const int slot = (pipe && !direct) ? (int)(t->flip % (uint32_t)mipt_tracer::kRing) : 0;
            if (pipe) t->flip++;
            hipStream_t S = set ? t->aux : t->stream;
            float* sample = direct ? t->blockSum : t->ring[slot];
            // pixel counters: a pipelined frame with a fold uses the block of its ring slot, which that fold clears again (the
            // slot's next frame already waits for it), so the 256-byte clear per frame -- a fill kernel that queues behind the
            // persistent kernel's resident blocks -- is only issued for a block that was left standing
            const bool ringCounter = pipe && !direct;
            uint32_t* counts = ringCounter ? t->ringCounts + slot * 64 : (set ? t->counts2 : t->counts);
            if (pipe) t->pipeBusy = true;
            const bool zeroAfter = pipe && !direct && !t->keepSample;
            if (!direct) {
                t->lastSlot = slot;
                if (pipe && t->foldValid[slot]) HIPCHK(t, hipStreamWaitEvent(S, t->foldEv[slot], 0));      // the slot's previous frame has been folded
                if (t->slotDirty[slot]) HIPCHK(t, hipMemsetAsync(sample, 0, t->slotDirty[slot] * 16, S));
                t->slotDirty[slot] = zeroAfter ? 0 : npix;
            }
            t->sampleConsumed = zeroAfter || direct;
            if (t->countPixelRays) HIPCHK(t, hipMemsetAsync(t->pixelRays, 0, (size_t)npix * 4, S));
            LevelArgs A;
            A.S = sc->dev; A.F = make_frame(*st, frame);
            A.sample = sample; A.pixelRays = t->countPixelRays ? t->pixelRays : nullptr; A.stats = t->stats;
            hipEvent_t ka = take_event(t), kb = take_event(t);      // integrator kernel time of this frame (they overlap across frames when pipelined)
            if (!ka || !kb) { give_event(t, ka); give_event(t, kb); return fail(t, MIPT_E_HIP, "hipEventCreate failed"); }
            t->pending.emplace_back(ka, kb);                        // from here on resolve_events returns them to the pool
            auto level_events = [&](mipt_tracer::LevelEv& le, hipStream_t on) -> bool {
                le.a = take_event(t); le.b = take_event(t);
                if (!le.a || !le.b) { give_event(t, le.a); give_event(t, le.b); le.a = le.b = nullptr; return false; }
                t->levelPending.push_back(le);
                return hipEventRecord(le.a, on) == hipSuccess;
            };
            if (t->rngMode == MIPT_RNG_DFS) {
                DfsArgs DA;
                DA.S = sc->dev; DA.F = A.F; DA.sample = A.sample; DA.pixelRays = A.pixelRays; DA.stats = t->stats; DA.pix0 = 0; DA.npix = npix;
                mipt_tracer::LevelEv le{ nullptr, nullptr, 0 };
                const bool prof = t->profileLevels && level_events(le, S);
                HIPCHK(t, hipEventRecord(ka, S));
                launch_dfs(DA, 0, S);
                HIPCHK(t, hipEventRecord(kb, S));
                if (prof) HIPCHK(t, hipEventRecord(le.b, S));
                t->launches++;
            } else if (t->integrator == 0 && D >= 1) {
                if (!ringCounter || !t->ringCountsClean[slot]) HIPCHK(t, hipMemsetAsync(counts, 0, 64 * sizeof(uint32_t), S));
                if (ringCounter) t->ringCountsClean[slot] = false;      // in use from here on; clean again once the fold that clears them is queued
                StackArgs SA;
                SA.S = sc->dev; SA.F = A.F; SA.stack = set ? t->wstack2 : t->wstack; SA.cap = wcap; SA.refill = (uint32_t)t->refillLanes;
                // min_descend_lanes auto: 8 for an LDS-resident scene, 24 otherwise (measured optima, profiles/r1/exp_min_descend.log)
                const bool ldsScene = wavestack_scene_in_lds(sc->dev);
                SA.minDescend = (uint32_t)(t->minDescend > 0 ? t->minDescend : (ldsScene ? 8 : 24));
            SA.pixCounter = counts; SA.nee = (uint32_t)t->estimator;
                // xcd_bands auto: on when nodes + references exceed one XCD's 4 MiB L2 or the frame has 4 Mi pixels or more.
                // Eight counters also spread the claims (one word sustains ~90 atomics/us, MI355X_MICROARCH.md; frames of cheap
                // pixels come close to that), which fits the gain growing as the pixels get cheaper: 4K grids
                // of 1.8 k ... 10 M triangles +32 ... +4 %, Cornell / Default at 4K +0.5 %; at 1080p the cache-resident
                // scenes lose to a longer tail (Cornell -2 %, Default -0.5 %) and keep one counter (profiles/r1/exp_xcd_bands.log)
                const size_t workingSet = (size_t)sc->dev.nNodes * 64 + (size_t)sc->dev.nRefs * 48;
                SA.xcdBands = (uint32_t)(t->xcdBands >= 0 ? t->xcdBands : ((workingSet > (4u << 20) || npixPart >= (1u << 22)) ? 1 : 0));
                SA.sample = A.sample; SA.pixelRays = A.pixelRays; SA.stats = t->stats; SA.pix0 = 0; SA.npix = npixPart; SA.band = band;
                SA.alphaOne = 0u;       // (the fold writes alpha = 1, also for a block of direct sum-mode frames)
                mipt_tracer::LevelEv le{ nullptr, nullptr, 0 };
                const bool prof = t->profileLevels && level_events(le, S);
                HIPCHK(t, hipEventRecord(ka, S));
                if (npixPart) launch_wavestack(SA, std::min<int>((int)((npixPart + 255) / 256), persistentBlocks), S);
                HIPCHK(t, hipEventRecord(kb, S));
                if (prof) HIPCHK(t, hipEventRecord(le.b, S));
                t->launches++;
            } else {
                HIPCHK(t, hipEventRecord(ka, S));
                for (uint32_t p0 = 0; p0 < npix; p0 += tile) {
                    const uint32_t np = std::min(tile, npix - p0);
                    HIPCHK(t, hipMemsetAsync(t->counts, 0, 64 * sizeof(uint32_t), S));
                    A.pix0 = p0; A.npix = np;
                    for (uint32_t k = 0; k < D; k++) {
                        A.level = k;
                        A.qin = t->q[k & 1]; A.qout = t->q[(k + 1) & 1];
                        A.inCount = t->counts + k; A.outCount = t->counts + k + 1;
                        int blocks = k == 0 ? std::min<int>((int)((np + 255) / 256), persistentBlocks) : persistentBlocks;
                        mipt_tracer::LevelEv le{ nullptr, nullptr, (int)k };
                        const bool prof = t->profileLevels && level_events(le, S);
                        launch_level(A, k == 0, blocks, S);
                        if (prof) HIPCHK(t, hipEventRecord(le.b, S));
                        t->launches++;
                    }
                }
                HIPCHK(t, hipEventRecord(kb, S));
            }
            if (!direct) {
                // fold into the caller's buffers, in frame order
                hipStream_t F = pipe ? t->fold : S;
                if (pipe) { HIPCHK(t, hipEventRecord(t->kernEv[slot], S)); HIPCHK(t, hipStreamWaitEvent(F, t->kernEv[slot], 0)); }
                const float weight = (float)(1.0 / (frame + 1.0));       // CUDATracerProg.cu:878
                mipt_tracer::LevelEv ae{ nullptr, nullptr, -1 };
                const bool prof = t->profileLevels && level_events(ae, F);
                launch_accumulate(sample, acc_dev, ldr_dev, weight, sum_mode, zeroAfter ? 1 : 0, (ringCounter && t->rngMode == MIPT_RNG_NODE && t->integrator == 0 && D >= 1) ? counts : nullptr, band, 0, npixPart, F);
                if (prof) HIPCHK(t, hipEventRecord(ae.b, F));
                if (pipe) { HIPCHK(t, hipEventRecord(t->foldEv[slot], F)); t->foldValid[slot] = true; }
                if (ringCounter && t->rngMode == MIPT_RNG_NODE && t->integrator == 0 && D >= 1) t->ringCountsClean[slot] = true;
            }
            if (direct && ++t->blockFrames >= mipt_tracer::kBlockFrames) flush_block(t);
            t->frames++;
        }
        HIPCHK(t, hipGetLastError());
        bookkeeping.ok = true;
        if (t->pending.size() > 256 || t->levelPending.size() > 4096) resolve_events(t);
        return MIPT_OK;
    });
}

int mipt_trace(mipt_tracer* t, mipt_scene* sc, const mipt_settings* st, float* acc_dev, unsigned char* ldr_dev) {
    if (!st) return fail(t, MIPT_E_INVALID, "null settings");
    return mipt_render(t, sc, st, st->frameID, 1, 1, 0, acc_dev, ldr_dev);
}

// mipt_trace for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): same contract and same kernels, but every work buffer -- pixel counter, sample buffer, wave stacks --
// lives in this GPU's `scratch` array, reached through the pointer mipt_create obtained.  No hipMalloc / hipFree /
// event creation on this path; everything is queued on the tracer's stream.  Limits: viewportWidth * viewportHeight
// <= 65536, maxDepth <= 8, estimator 0 or 2, integrator 0, rng_mode node, no tile partition.  Calls on one GPU share
// the array: the parts that must start at zero are cleared per call, and calls that overlap on one GPU (two tracers,
// or two streams) race.
int mipt_trace_static(mipt_tracer* t, mipt_scene* sc, const mipt_settings* st, float* acc_dev, unsigned char* ldr_dev) {
    if (!t || !sc || !st || !acc_dev) return fail(t, MIPT_E_INVALID, "null argument");
    if (sc->t != t) return fail(t, MIPT_E_INVALID, "scene belongs to another tracer");
    const uint32_t W = st->viewportWidth, H = st->viewportHeight, D = st->maxDepth;
    if (!W || !H) return fail(t, MIPT_E_INVALID, "empty viewport");
    if ((uint64_t)W * H > kStaticMaxPixels) return fail(t, MIPT_E_LIMIT, "mipt_trace_static: viewport exceeds 65536 pixels (the static scratch holds one sample buffer of that size)");
    if (D > kStaticMaxDepth) return fail(t, MIPT_E_LIMIT, "mipt_trace_static: maxDepth exceeds 8 (the static scratch holds 256 wave stacks of 256 * 8 + 256 rays)");
    if (t->rngMode != MIPT_RNG_NODE || t->integrator != 0 || t->tileParts > 1 || t->countPixelRays || (t->estimator != 0 && t->estimator != 2))
        return fail(t, MIPT_E_INVALID, "mipt_trace_static: needs integrator 0, rng_mode node, estimator 0 or 2, no tile partition, no per-pixel ray counting");
    if (!t->staticScratch) return fail(t, MIPT_E_HIP, "mipt_trace_static: no static scratch on this device");
    if (t->pipeBusy) {
        // Joining pipelined frames means event waits between the tracer's internal streams and t->stream: under stream
        // capture that would tie the capture to work outside it.  (mipt_set_stream joins before it switches, so a caller
        // capturing its own stream never gets here; the check is for a stream handed over some other way.)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(t->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(t, MIPT_E_INVALID, "mipt_trace_static: pipelined frames are still in flight and the stream is being captured; call mipt_synchronize before the capture begins");
    }
    if (use(t)) return MIPT_E_HIP;
    const uint32_t npix = W * H;
    unsigned char* base = (unsigned char*)t->staticScratch;
    uint32_t* counters = (uint32_t*)base;
    float* sample = (float*)(base + kStaticSampleOffset);
    const uint32_t wcap = 256u * std::max<uint32_t>(D, 1u) + 256u;
    const size_t arr = (size_t)kStaticBlocks * 4 * wcap * sizeof(float4);
    HIPCHK(t, hipMemsetAsync(counters, 0, 256, t->stream));
    HIPCHK(t, hipMemsetAsync(sample, 0, (size_t)npix * 16, t->stream));
    const FrameDev F = make_frame(*st, st->frameID);
    if (D >= 1) {
        StackArgs SA;
        SA.S = sc->dev; SA.F = F;
        SA.stack.a = (float4*)(base + kStaticStackOffset); SA.stack.b = (float4*)(base + kStaticStackOffset + arr); SA.stack.c = (float4*)(base + kStaticStackOffset + 2 * arr);
        SA.cap = wcap; SA.refill = (uint32_t)t->refillLanes;
        SA.minDescend = (uint32_t)(t->minDescend > 0 ? t->minDescend : (wavestack_scene_in_lds(sc->dev) ? 8 : 24));
        SA.pixCounter = counters; SA.nee = (uint32_t)t->estimator; SA.xcdBands = 0;
        SA.sample = sample; SA.pixelRays = nullptr; SA.stats = t->stats; SA.pix0 = 0; SA.npix = npix; SA.band = Band{ 0u, 1u, 0u }; SA.alphaOne = 0;
        launch_wavestack(SA, std::min<int>((int)((npix + 255) / 256), kStaticBlocks), t->stream);
        t->launches++;
    }
    const float weight = (float)(1.0 / (st->frameID + 1.0));       // CUDATracerProg.cu:878
    launch_accumulate(sample, acc_dev, ldr_dev, weight, 0, 0, nullptr, Band{ 0u, 1u, 0u }, 0, npix, t->stream);
    HIPCHK(t, hipGetLastError());
    t->sampleConsumed = true;          // mipt_last_sample reads the tracer's own sample buffers, not the scratch
    t->frames++;
    return MIPT_OK;
}

int mipt_tonemap(mipt_tracer* t, const float* acc_dev, float scale, uint32_t n_pixels, unsigned char* ldr_dev) {
    return guarded(t, [&]() -> int {
        if (!t || !acc_dev || !ldr_dev) return fail(t, MIPT_E_INVALID, "null argument");
        if (use(t)) return MIPT_E_HIP;
        launch_tonemap(acc_dev, scale, n_pixels, ldr_dev, t->stream);
        HIPCHK(t, hipGetLastError());
        return MIPT_OK;
    });
}

int mipt_last_sample(mipt_tracer* t, float* rgb_host, uint32_t* rays_host, uint32_t n_pixels) {
    return guarded(t, [&]() -> int {
        if (!t) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        if (t->sampleConsumed) return fail(t, MIPT_E_INVALID, "the last frame's sample buffer was consumed by its fold (or summed straight into the accumulation buffer): set option keep_sample = 1 before rendering");
        if (n_pixels > (t->lastSlot ? t->ringPixels : t->samplePixels)) return fail(t, MIPT_E_INVALID, "no sample of that size");
        HIPCHK(t, hipStreamSynchronize(t->stream));
        if (rgb_host) {
            std::vector<float> tmp((size_t)n_pixels * 4);
            HIPCHK(t, hipMemcpy(tmp.data(), t->lastSlot ? t->ring[t->lastSlot] : t->sample, tmp.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n_pixels; i++) { rgb_host[i * 3] = tmp[i * 4]; rgb_host[i * 3 + 1] = tmp[i * 4 + 1]; rgb_host[i * 3 + 2] = tmp[i * 4 + 2]; }
        }
        if (rays_host) {
            if (!t->pixelRays || t->pixelRaysCap < n_pixels) return fail(t, MIPT_E_INVALID, "count_pixel_rays was not enabled");
            HIPCHK(t, hipMemcpy(rays_host, t->pixelRays, (size_t)n_pixels * 4, hipMemcpyDeviceToHost));
        }
        return MIPT_OK;
    });
}

int mipt_trace_rays(mipt_tracer* t, mipt_scene* sc, const float* o, const float* d, uint32_t n, mipt_hit* hits) {
    return guarded(t, [&]() -> int {
        if (!t || !sc || (n && (!o || !d || !hits))) return fail(t, MIPT_E_INVALID, "null argument");
        if (sc->t != t) return fail(t, MIPT_E_INVALID, "scene belongs to another tracer");
        if (use(t)) return MIPT_E_HIP;
        launch_trace_rays(sc->dev, o, d, n, hits, t->stream);
        HIPCHK(t, hipGetLastError());
        return MIPT_OK;
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Device buffer layer (CUDABuffer's lazy host/device mirror; first touch is zero-filled, CUDABuffer.cpp:114-118,140-145)
// ---------------------------------------------------------------------------------------------------------------
int mipt_buffer_create(mipt_tracer* t, size_t size, mipt_buffer** out) {
    return guarded(t, [&]() -> int {
        if (!t || !out) return MIPT_E_INVALID;
        mipt_buffer* b = new mipt_buffer(); b->t = t; b->size = size; *out = b;
        return MIPT_OK;
    });
}
void mipt_buffer_destroy(mipt_buffer* b) {
    if (!b) return;
    hipSetDevice(b->t->device);
    if (b->cpu && b->ownCpu) hipHostFree(b->cpu);
    if (b->gpu && b->ownGpu) hipFree(b->gpu);
    delete b;
}
// to_cpu / to_gpu of the reference (CUDABuffer.cpp:111-161).  Only the branches that move or allocate anything order
// themselves after the frames in flight (use() -> join_aux); a call that finds the head where it is wanted touches no
// stream, so reading the settings block before every Trace (CUDATracerProg.cu:942) does not serialise the frame pipeline.
static bool buf_to_cpu(mipt_buffer* b) {
    size_t sz = b->size ? b->size : 1;
    switch (b->head) {
        case MIPT_HEAD_UNINITIALIZED:
            if (use_device(b->t)) return false;
            if (hipHostMalloc(&b->cpu, sz, hipHostMallocDefault) != hipSuccess) return false;
            b->ownCpu = true;
            memset(b->cpu, 0, sz); b->head = MIPT_HEAD_AT_CPU; break;
        case MIPT_HEAD_AT_GPU:
            if (use(b->t)) return false;
            if (!b->cpu) { if (hipHostMalloc(&b->cpu, sz, hipHostMallocDefault) != hipSuccess) return false; b->ownCpu = true; }
            // blocking copy ordered after the tracer's stream, as the reference's default-stream cudaMemcpy is
            if (hipStreamSynchronize(b->t->stream) != hipSuccess) return false;
            if (hipMemcpy(b->cpu, b->gpu, b->size, hipMemcpyDeviceToHost) != hipSuccess) return false;
            b->head = MIPT_HEAD_SYNCED; break;
        default: break;
    }
    return true;
}
static bool buf_to_gpu(mipt_buffer* b) {
    size_t sz = b->size ? b->size : 1;
    switch (b->head) {
        case MIPT_HEAD_UNINITIALIZED:
            if (use(b->t)) return false;
            if (hipMalloc(&b->gpu, sz) != hipSuccess) return false;
            b->ownGpu = true;
            if (hipMemsetAsync(b->gpu, 0, sz, b->t->stream) != hipSuccess) return false;
            if (hipStreamSynchronize(b->t->stream) != hipSuccess) return false;
            b->head = MIPT_HEAD_AT_GPU; break;
        case MIPT_HEAD_AT_CPU:
            if (use(b->t)) return false;
            if (!b->gpu) { if (hipMalloc(&b->gpu, sz) != hipSuccess) return false; b->ownGpu = true; }
            if (hipStreamSynchronize(b->t->stream) != hipSuccess) return false;
            if (hipMemcpy(b->gpu, b->cpu, b->size, hipMemcpyHostToDevice) != hipSuccess) return false;
            b->head = MIPT_HEAD_SYNCED; break;
        default: break;
    }
    return true;
}
const void* mipt_buffer_cpu_data(mipt_buffer* b) { if (!b || !buf_to_cpu(b)) return nullptr; return b->cpu; }
const void* mipt_buffer_gpu_data(mipt_buffer* b) { if (!b || !buf_to_gpu(b)) return nullptr; return b->gpu; }
void* mipt_buffer_mutable_cpu_data(mipt_buffer* b) { if (!b || !buf_to_cpu(b)) return nullptr; b->head = MIPT_HEAD_AT_CPU; return b->cpu; }
void* mipt_buffer_mutable_gpu_data(mipt_buffer* b) { if (!b || !buf_to_gpu(b)) return nullptr; b->head = MIPT_HEAD_AT_GPU; return b->gpu; }
int mipt_buffer_head(const mipt_buffer* b) { return b ? b->head : MIPT_E_INVALID; }
size_t mipt_buffer_size(const mipt_buffer* b) { return b ? b->size : 0; }
// CUDABuffer::set_cpu_data / set_gpu_data (CUDABuffer.cpp:166-191): the buffer adopts caller-owned memory of at least
// size() bytes without taking ownership; the head moves to that side.
int mipt_buffer_set_cpu_data(mipt_buffer* b, void* data) {
    if (!b || !data) return MIPT_E_INVALID;
    if (use_device(b->t)) return MIPT_E_HIP;
    if (b->cpu && b->ownCpu) hipHostFree(b->cpu);
    b->cpu = data; b->ownCpu = false; b->head = MIPT_HEAD_AT_CPU;
    return MIPT_OK;
}
int mipt_buffer_set_gpu_data(mipt_buffer* b, void* data_dev) {
    if (!b || !data_dev) return MIPT_E_INVALID;
    if (use(b->t)) return MIPT_E_HIP;                 // frames in flight may still be using the memory let go of here
    if (b->gpu && b->ownGpu) { hipStreamSynchronize(b->t->stream); hipFree(b->gpu); }
    b->gpu = data_dev; b->ownGpu = false; b->head = MIPT_HEAD_AT_GPU;
    return MIPT_OK;
}
// CUDABuffer::async_gpu_push(stream) (CUDABuffer.cpp:208-224): head must be at the CPU; the copy is queued on the
// caller's stream (NULL: the tracer's) and the caller synchronises on it before using the device side.
int mipt_buffer_async_gpu_push_on(mipt_buffer* b, void* hip_stream) {
    if (!b || b->head != MIPT_HEAD_AT_CPU) return MIPT_E_INVALID;
    mipt_tracer* t = b->t;
    if (use(t)) return MIPT_E_HIP;
    if (!b->gpu) { HIPCHK(t, hipMalloc(&b->gpu, b->size ? b->size : 1)); b->ownGpu = true; }
    HIPCHK(t, hipMemcpyAsync(b->gpu, b->cpu, b->size, hipMemcpyHostToDevice, hip_stream ? (hipStream_t)hip_stream : t->stream));
    b->head = MIPT_HEAD_SYNCED;
    return MIPT_OK;
}
int mipt_buffer_async_gpu_push(mipt_buffer* b) { return mipt_buffer_async_gpu_push_on(b, nullptr); }

int mipt_malloc(mipt_tracer* t, size_t size, void** dev_out) {
    return guarded(t, [&]() -> int {
        if (!t || !dev_out) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        HIPCHK(t, hipMalloc(dev_out, size ? size : 1));
        return MIPT_OK;
    });
}
int mipt_free(mipt_tracer* t, void* dev) { if (!t) return MIPT_E_INVALID; if (use(t)) return MIPT_E_HIP; if (dev) HIPCHK(t, hipFree(dev)); return MIPT_OK; }
int mipt_upload(mipt_tracer* t, void* dev, const void* host, size_t size) {
    return guarded(t, [&]() -> int {
        if (!t || (size && (!dev || !host))) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        HIPCHK(t, hipStreamSynchronize(t->stream));
        if (size) HIPCHK(t, hipMemcpy(dev, host, size, hipMemcpyHostToDevice));
        return MIPT_OK;
    });
}
int mipt_download(mipt_tracer* t, void* host, const void* dev, size_t size) {
    return guarded(t, [&]() -> int {
        if (!t || (size && (!dev || !host))) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        HIPCHK(t, hipStreamSynchronize(t->stream));
        if (size) HIPCHK(t, hipMemcpy(host, dev, size, hipMemcpyDeviceToHost));
        return MIPT_OK;
    });
}
int mipt_memset(mipt_tracer* t, void* dev, int value, size_t size) {
    return guarded(t, [&]() -> int {
        if (!t || (size && !dev)) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        if (size) HIPCHK(t, hipMemsetAsync(dev, value, size, t->stream));
        return MIPT_OK;
    });
}

int mipt_test_math(mipt_tracer* t, int fn, const float* a, const float* b, uint32_t n, float* out) {
    return guarded(t, [&]() -> int {
        if (!t || !a || !out) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        float *da = nullptr, *db = nullptr, *dout = nullptr;
        struct Cleanup { float*& a; float*& b; float*& c; ~Cleanup() { if (a) hipFree(a); if (b) hipFree(b); if (c) hipFree(c); } } cleanup{ da, db, dout };
        size_t bytes = (size_t)std::max<uint32_t>(n, 1) * 4;
        HIPCHK(t, hipMalloc((void**)&da, bytes)); HIPCHK(t, hipMalloc((void**)&db, bytes)); HIPCHK(t, hipMalloc((void**)&dout, bytes));
        HIPCHK(t, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
        if (b) HIPCHK(t, hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice)); else HIPCHK(t, hipMemset(db, 0, bytes));
        launch_test_math(fn, da, db, n, dout, t->stream);
        HIPCHK(t, hipStreamSynchronize(t->stream));
        HIPCHK(t, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
        return MIPT_OK;
    });
}

// Test support for option "wave_stack_entries": out[0] = bytes of the wave-stack arrays beyond the waves' regions,
// out[1] = how many of them no longer hold the paint mipt_render put there (must be 0: a wave never writes past its stack),
// out[2] = shade batches cut short or put off for lack of stack room since the last mipt_reset_counters (any capacity).
int mipt_test_stack_guard(mipt_tracer* t, uint64_t out[3]) {
    return guarded(t, [&]() -> int {
        if (!t || !out) return MIPT_E_INVALID;
        if (use(t)) return MIPT_E_HIP;
        HIPCHK(t, hipStreamSynchronize(t->stream));
        out[0] = out[1] = out[2] = 0;
        unsigned long long cut = 0;
        HIPCHK(t, hipMemcpy(&cut, t->stats + 36, sizeof cut, hipMemcpyDeviceToHost));
        out[2] = cut;
        if (!t->waveStackEntries) return MIPT_OK;          // no paint, nothing to check: only the counter
        std::vector<unsigned char> h;
        for (Queue* q : { &t->wstack, &t->wstack2 }) {
            const size_t n = q == &t->wstack ? t->wstackEntries : t->wstack2Entries;
            if (!n || !q->a || n <= t->guardFrom) continue;
            const size_t bytes = (n - t->guardFrom) * sizeof(float4);
            h.resize(bytes);
            for (float4* arr : { q->a, q->b, q->c }) {
                HIPCHK(t, hipMemcpy(h.data(), arr + t->guardFrom, bytes, hipMemcpyDeviceToHost));
                out[0] += bytes;
                for (unsigned char c : h) if (c != 0xA5) out[1]++;
            }
        }
        return MIPT_OK;
    });
}

// VALU issue calibration (csrc/microbench.hip): `iters` x 64 independent v_fma_f32 per wave on numCU x blocks_per_cu
// blocks of 256 threads; median per-wave shader-clock cycles and 100 MHz ticks of the loop, HIP-event time of the launch.
int mipt_test_issue(mipt_tracer* t, int packed, uint32_t blocks_per_cu, uint32_t iters, mipt_issue_result* out) {
    return guarded(t, [&]() -> int {
        if (!t || !out || blocks_per_cu < 1 || blocks_per_cu > 8 || iters < 1 || iters > (1u << 24)) return fail(t, MIPT_E_INVALID, "test_issue: blocks_per_cu 1..8, iters 1..2^24");
        if (use(t)) return MIPT_E_HIP;
        const int blocks = t->numCU * (int)blocks_per_cu;
        const size_t waves = (size_t)blocks * 4;
        unsigned long long* stamps = nullptr; float* sink = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        struct Cleanup {        // every failure path below returns through HIPCHK: nothing may stay behind
            mipt_tracer* t; unsigned long long*& stamps; float*& sink; hipEvent_t& a; hipEvent_t& b;
            ~Cleanup() { if (stamps) hipFree(stamps); if (sink) hipFree(sink); give_event(t, a); give_event(t, b); }
        } cleanup{ t, stamps, sink, a, b };
        HIPCHK(t, hipMalloc((void**)&stamps, waves * 2 * sizeof(unsigned long long)));
        HIPCHK(t, hipMalloc((void**)&sink, 256 * sizeof(float)));
        a = take_event(t); b = take_event(t);          // from the tracer's pool, like the frame loop's
        if (!a || !b) return fail(t, MIPT_E_HIP, "hipEventCreate failed");
        launch_test_issue(packed, iters, blocks, stamps, sink, t->stream);            // warm-up: code object load, clocks
        HIPCHK(t, hipEventRecord(a, t->stream));
        launch_test_issue(packed, iters, blocks, stamps, sink, t->stream);
        HIPCHK(t, hipEventRecord(b, t->stream));
        HIPCHK(t, hipStreamSynchronize(t->stream));
        float ms = 0.f; HIPCHK(t, hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h(waves * 2);
        HIPCHK(t, hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> cyc(waves), rt(waves);
        std::vector<double> ghz(waves);
        for (size_t w = 0; w < waves; w++) { cyc[w] = h[2 * w]; rt[w] = h[2 * w + 1]; ghz[w] = rt[w] ? (double)cyc[w] / ((double)rt[w] * 10.0) : 0.0; }
        std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end()); std::sort(ghz.begin(), ghz.end());
        out->clock_ghz = ghz[waves / 2];
        out->ms = ms; out->waves = (uint32_t)waves; out->simds = (uint32_t)t->numCU * 4u;
        out->wave_instructions = (uint64_t)waves * iters * 64ull;
        out->median_cycles = (double)cyc[waves / 2]; out->max_cycles = (double)cyc[waves - 1];
        out->median_realtime_ticks = (double)rt[waves / 2];
        return MIPT_OK;
    });
}

// Exhaustive comparison of the integrator's short reciprocal (device_math.h rcp_newton<steps>) with the compiler's IEEE
// division over every float of magnitude 2^-100 .. 2^100.  out[0] compared, out[1] differing, out[2] first differing input
// (bits, or 0), out[3] differing by more than one unit in the last place.  steps = 4: the same for the sampler's short square
// root (device_math.h sqrt_unit) against sqrtf, over zero and every positive float in 2^-96 .. 2^100.
int mipt_test_rcp_sweep(mipt_tracer* t, int steps, uint64_t out[4]) {
    return guarded(t, [&]() -> int {
        if (!t || !out || steps < 0 || steps > 4) return fail(t, MIPT_E_INVALID, "test_rcp_sweep: steps 0..3, or 4 for the square root");
        if (use(t)) return MIPT_E_HIP;
        unsigned long long* d = nullptr;
        struct Cleanup { unsigned long long*& d; ~Cleanup() { if (d) hipFree(d); } } cleanup{ d };
        HIPCHK(t, hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
        const unsigned long long init[4] = { 0ull, 0ull, ~0ull, 0ull };
        HIPCHK(t, hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
        launch_test_rcp_sweep(steps, t->numCU * 8, d, t->stream);
        HIPCHK(t, hipStreamSynchronize(t->stream));
        unsigned long long h[4];
        HIPCHK(t, hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        out[0] = h[0]; out[1] = h[1]; out[2] = h[2] == ~0ull ? 0ull : (h[2] & 0xFFFFFFFFull); out[3] = h[3];
        return MIPT_OK;
    });
}

}  // extern "C"
