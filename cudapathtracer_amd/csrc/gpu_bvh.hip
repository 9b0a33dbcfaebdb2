// gpu_bvh.hip -- linear BVH built on the GPU, emitted in the flat layout of bvh_builder.h.  See gpu_bvh.h.
//
// Passes (n triangles, all on one stream):
//   1 k_tri_boxes     box of every triangle; bounds of the centroids (wave reduction + one atomic per wave and component)
//   2 k_morton        63-bit Morton code of the centroid (21 bits per axis), value = triangle index
//   3 rocprim         radix_sort_pairs on the 63 key bits (stable: equal codes keep index order)
//   4 k_radix_tree    Karras 2012: inner node i of the binary radix tree over the sorted codes (ties broken by position),
//                     one thread per inner node, no communication
//   5 k_fit           bottom-up: a thread per triangle climbs; the second arrival at a node unions the children's boxes
//                     (no waiting: the first arrival leaves; visibility by __threadfence either side of the counter)
//   6 k_keep + scan   inner nodes that cover more than maxLeafSize triangles are kept and renumbered (exclusive scan);
//     k_emit2         the others become leaf references [first, count] into the sorted order -- the reference array IS the
//                     sorted index array -- and kept nodes are written as BVH2 records (child boxes + child references)
//   7 k_wide_count / scan / k_wide_emit, once per level: the host Collapser's rule (open the inner child with the largest
//                     area until four children), child indices from a scan so the numbering is deterministic, nodes
//                     quantised by the shared quantise_node
// Nothing in here spins or waits on another thread, every loop is bounded by the key width, and every index that comes
// out of device memory is checked on the host (validate_bvh) before a traversal kernel sees it.
#include "gpu_bvh.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "bvh_quantise.h"

namespace mipt {
namespace {

constexpr int kB = 256;

__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float ord2f(uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f; }

__device__ inline float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ inline float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// ---- 1 ----
__global__ void __launch_bounds__(kB) k_tri_boxes(const float* __restrict__ tris, uint32_t n, QBox* __restrict__ leafBox, uint32_t* __restrict__ cb /* 3 min, 3 max as ordered uints */) {
    float cmn[3] = { INFINITY, INFINITY, INFINITY }, cmx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        const float* T = tris + (size_t)i * 9;
        QBox b;
        for (int a = 0; a < 3; a++) {
            b.mn[a] = fminf(fminf(T[a], T[3 + a]), T[6 + a]);
            b.mx[a] = fmaxf(fmaxf(T[a], T[3 + a]), T[6 + a]);
            const float c = 0.5f * b.mn[a] + 0.5f * b.mx[a];
            cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c);
        }
        leafBox[i] = b;
    }
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(cmn[a]), hi = wave_max(cmx[a]);
        if ((threadIdx.x & 63) == 0) { atomicMin(cb + a, f2ord(lo)); atomicMax(cb + 3 + a, f2ord(hi)); }
    }
}

// ---- 2 ----
__device__ inline unsigned long long spread21(uint32_t v) {
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
struct MortonFrame { float org[3], inv[3]; };     // cell = (c - org) * inv, inv = 2^21 / extent (0 for a flat axis)

__global__ void __launch_bounds__(kB) k_morton(const QBox* __restrict__ leafBox, uint32_t n, MortonFrame F, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const QBox b = leafBox[i];
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float c = 0.5f * b.mn[a] + 0.5f * b.mx[a];
        float x = (c - F.org[a]) * F.inv[a];
        x = fminf(fmaxf(x, 0.f), 2097151.f);
        q[a] = (uint32_t)x;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// ---- 4 ----  (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012, section 4)
// common prefix of the keys at positions i and j; equal keys are told apart by their positions; -1 outside the array
__device__ inline int delta(const unsigned long long* __restrict__ k, int n, int i, unsigned long long ki, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = ki ^ k[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}
// children are stored as: inner node index >= 0, or ~(leaf position) < 0
__global__ void __launch_bounds__(kB) k_radix_tree(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ child, int2* __restrict__ range,
                                                    int* __restrict__ parentInner, int* __restrict__ parentLeaf) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const unsigned long long ki = keys[i];
    const int d = delta(keys, n, i, ki, i + 1) - delta(keys, n, i, ki, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, ki, i - d);
    int lmax = 2;
    while (delta(keys, n, i, ki, i + lmax * d) > dmin && lmax < (1 << 30)) lmax <<= 1;        // (bounded: at most log2 n doublings)
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, ki, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, ki, j);
    int s = 0;
    // steps ceil(l/2), ceil(l/4), ..., 1 (Karras' do-while; for l == 1 the single step cannot move s and is skipped)
    for (int t = l; t > 1;) { t = (t + 1) >> 1; if (delta(keys, n, i, ki, i + (s + t) * d) > dnode) s += t; }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? ~gamma : gamma, right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    range[i] = make_int2(lo, hi);
    if (left >= 0) parentInner[left] = i; else parentLeaf[~left] = i;
    if (right >= 0) parentInner[right] = i; else parentLeaf[~right] = i;
    if (i == 0) parentInner[0] = -1;
}

// ---- 5 ----
__device__ inline QBox box_union(const QBox& a, const QBox& b) {
    QBox r;
    for (int k = 0; k < 3; k++) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return r;
}
__global__ void __launch_bounds__(kB) k_fit(int n, const uint32_t* __restrict__ sortedTri, const QBox* __restrict__ leafBox, const int2* __restrict__ child,
                                             const int* __restrict__ parentInner, const int* __restrict__ parentLeaf, QBox* nodeBox, uint32_t* arrived) {
    const int k = blockIdx.x * kB + threadIdx.x;
    if (k >= n) return;
    int node = parentLeaf[k];
    for (int guard = 0; node >= 0 && guard < 128; guard++) {          // (a radix tree over 63 + 32 bits is at most 95 levels deep)
        __threadfence();                                               // what this thread wrote below is visible before it is counted
        if (atomicAdd(arrived + node, 1u) == 0u) return;               // first at this node: the sibling's side is not finished, leave
        __threadfence();
        const int2 c = child[node];
        const QBox a = c.x >= 0 ? nodeBox[c.x] : leafBox[sortedTri[~c.x]];
        const QBox b = c.y >= 0 ? nodeBox[c.y] : leafBox[sortedTri[~c.y]];
        nodeBox[node] = box_union(a, b);
        node = parentInner[node];
    }
}

// ---- 6 ----
__global__ void __launch_bounds__(kB) k_keep(int nInner, const int2* __restrict__ range, uint32_t maxLeaf, uint32_t* __restrict__ keep) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nInner) return;
    const int2 r = range[i];
    keep[i] = (uint32_t)(r.y - r.x + 1) > maxLeaf ? 1u : 0u;
}
__device__ inline uint32_t leaf_ref(uint32_t first, uint32_t count) { return 0x80000000u | (count << 27) | first; }

__global__ void __launch_bounds__(kB) k_emit2(int nInner, const int2* __restrict__ child, const int2* __restrict__ range, const uint32_t* __restrict__ keep,
                                               const uint32_t* __restrict__ newIndex, const uint32_t* __restrict__ sortedTri, const QBox* __restrict__ leafBox,
                                               const QBox* __restrict__ nodeBox, float* __restrict__ nodes) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nInner || !keep[i]) return;
    float* P = nodes + (size_t)newIndex[i] * 16;
    const int2 c = child[i];
    const int cc[2] = { c.x, c.y };
    uint32_t refs[2];
    for (int s = 0; s < 2; s++) {
        QBox b;
        if (cc[s] < 0) { const uint32_t pos = (uint32_t)~cc[s]; b = leafBox[sortedTri[pos]]; refs[s] = leaf_ref(pos, 1u); }
        else {
            b = nodeBox[cc[s]];
            const int2 r = range[cc[s]];
            refs[s] = keep[cc[s]] ? newIndex[cc[s]] : leaf_ref((uint32_t)r.x, (uint32_t)(r.y - r.x + 1));
        }
        P[s * 6 + 0] = b.mn[0]; P[s * 6 + 1] = b.mn[1]; P[s * 6 + 2] = b.mn[2];
        P[s * 6 + 3] = b.mx[0]; P[s * 6 + 4] = b.mx[1]; P[s * 6 + 5] = b.mx[2];
    }
    P[12] = __uint_as_float(refs[0]); P[13] = __uint_as_float(refs[1]); P[14] = 0.f; P[15] = 0.f;
}

// ---- 7 ----  one 4-wide node per queue entry (x = BVH2 node it is rooted at, y = its own index in the 4-wide array)
struct WideKids { uint32_t ref[4]; QBox box[4]; int n; };
__device__ inline void gather_kids(const float* __restrict__ nodes2, uint32_t nNodes2, uint32_t node2, WideKids& K) {
    auto kid = [&](uint32_t nd, int which, int slot) {
        const float* P = nodes2 + (size_t)nd * 16;
        for (int a = 0; a < 3; a++) { K.box[slot].mn[a] = P[which * 6 + a]; K.box[slot].mx[a] = P[which * 6 + 3 + a]; }
        K.ref[slot] = __float_as_uint(P[12 + which]);
    };
    kid(node2, 0, 0); kid(node2, 1, 1); K.n = 2;
    while (K.n < 4) {                                  // open the inner child with the largest area (bvh_builder.cpp Collapser::build)
        int best = -1; float bestA = -1.f;
        for (int i = 0; i < K.n; i++) if (!(K.ref[i] & 0x80000000u)) { const float a = K.box[i].area(); if (a > bestA) { bestA = a; best = i; } }
        if (best < 0) break;
        const uint32_t nd = K.ref[best];
        if (nd >= nNodes2) break;                      // (cannot happen for k_emit2's output; never index past the array)
        kid(nd, 0, best); kid(nd, 1, K.n); K.n++;
    }
}
__global__ void __launch_bounds__(kB) k_wide_count(const float* __restrict__ nodes2, uint32_t nNodes2, const uint2* __restrict__ queue, uint32_t nq, uint32_t* __restrict__ cnt) {
    const uint32_t e = blockIdx.x * kB + threadIdx.x;
    if (e >= nq) return;
    WideKids K; gather_kids(nodes2, nNodes2, queue[e].x, K);
    uint32_t k = 0;
    for (int i = 0; i < K.n; i++) k += (K.ref[i] & 0x80000000u) ? 0u : 1u;
    cnt[e] = k;
}
__global__ void __launch_bounds__(kB) k_wide_emit(const float* __restrict__ nodes2, uint32_t nNodes2, const uint2* __restrict__ queue, uint32_t nq, const uint32_t* __restrict__ offset,
                                                   uint32_t nextBase, uint32_t capWide, float* __restrict__ nodes4, uint2* __restrict__ next, uint32_t* __restrict__ err) {
    const uint32_t e = blockIdx.x * kB + threadIdx.x;
    if (e >= nq) return;
    const uint2 me = queue[e];
    WideKids K; gather_kids(nodes2, nNodes2, me.x, K);
    uint32_t refs[4] = { 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u };      // kEmptyRef
    uint32_t o = offset[e];
    for (int i = 0; i < K.n; i++) {
        if (K.ref[i] & 0x80000000u) refs[i] = K.ref[i];
        else {
            const uint32_t idx = nextBase + o;
            if (idx >= capWide) { atomicOr(err, 2u); refs[i] = 0x80000000u; continue; }
            refs[i] = idx; next[o] = make_uint2(K.ref[i], idx); o++;
        }
    }
    if (me.y >= capWide) { atomicOr(err, 2u); return; }
    if (!quantise_node(nodes4 + (size_t)me.y * 16, K.box, K.n, refs)) atomicOr(err, 1u);
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class T> T* as() { return (T*)p; }
};
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

#define GB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return std::string("GPU BVH build: ") + #x + ": " + hipGetErrorString(e_); } while (0)

std::string build_bvh_gpu(const std::vector<TriPos>& tris, const BvhParams& p, BvhOut& out, hipStream_t stream, GpuBuildTimes* times) {
    const size_t nt = tris.size();
    if (nt < kGpuBuildMinTris) return "fewer than 64 triangles: use a host builder (bvhBuilder 0 or 1)";
    if (nt >= (1u << 26)) return "too many triangles for 27-bit reference indices";
    if (p.maxLeafSize < 1 || p.maxLeafSize > 15) return "maxLeafSize must be in [1,15]";
    if (p.nodeWidth != 0 && p.nodeWidth != 2 && p.nodeWidth != 4) return "nodeWidth must be 0 (auto), 2 or 4";
    const int n = (int)nt, nInner = n - 1;
    const uint32_t maxLeaf = p.maxLeafSize;
    GpuBuildTimes T;
    double t0 = now_ms();

    DevBuf dTris, dLeafBox, dCb, dKeys, dKeys2, dVals, dVals2, dChild, dRange, dParI, dParL, dNodeBox, dArrived, dKeep, dNewIdx, dTemp;
    if (!dTris.alloc(nt * 36) || !dLeafBox.alloc(nt * sizeof(QBox)) || !dCb.alloc(32) || !dKeys.alloc(nt * 8) || !dKeys2.alloc(nt * 8) ||
        !dVals.alloc(nt * 4) || !dVals2.alloc(nt * 4) || !dChild.alloc((size_t)nInner * 8) || !dRange.alloc((size_t)nInner * 8) ||
        !dParI.alloc((size_t)nInner * 4) || !dParL.alloc(nt * 4) || !dNodeBox.alloc((size_t)nInner * sizeof(QBox)) ||
        !dArrived.alloc((size_t)nInner * 4) || !dKeep.alloc((size_t)nInner * 4) || !dNewIdx.alloc((size_t)nInner * 4))
        return "GPU BVH build: out of device memory";
    static_assert(sizeof(TriPos) == 36, "TriPos is nine floats");
    GB_HIP(hipMemcpyAsync(dTris.p, tris.data(), nt * 36, hipMemcpyHostToDevice, stream));
    const uint32_t cbInit[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    GB_HIP(hipMemcpyAsync(dCb.p, cbInit, sizeof cbInit, hipMemcpyHostToDevice, stream));
    GB_HIP(hipStreamSynchronize(stream));
    T.uploadMs = (float)(now_ms() - t0); t0 = now_ms();

    const int blocksN = (n + kB - 1) / kB, blocksI = (nInner + kB - 1) / kB;
    hipLaunchKernelGGL(k_tri_boxes, dim3(blocksN < 2048 ? blocksN : 2048), dim3(kB), 0, stream, dTris.as<float>(), (uint32_t)n, dLeafBox.as<QBox>(), dCb.as<uint32_t>());
    uint32_t cb[6];
    GB_HIP(hipMemcpyAsync(cb, dCb.p, sizeof cb, hipMemcpyDeviceToHost, stream));
    GB_HIP(hipStreamSynchronize(stream));
    // cubic cells: one scale for the three axes (the largest extent of the centroids), so that a flat or elongated scene is
    // not cut into slabs along its short axes (measured on the 10 M-triangle height field: 3.3 -> see exp_gpu_build.log)
    MortonFrame F;
    float maxExt = 0.f;
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(cb[a]), hi = ord2f(cb[3 + a]);
        if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) return "GPU BVH build: triangle coordinates are not finite";
        F.org[a] = lo;
        maxExt = std::max(maxExt, hi - lo);
    }
    float inv = maxExt > 0.f ? 2097152.f / maxExt : 0.f;
    if (!std::isfinite(inv)) inv = 0.f;
    for (int a = 0; a < 3; a++) F.inv[a] = inv;
    hipLaunchKernelGGL(k_morton, dim3(blocksN), dim3(kB), 0, stream, dLeafBox.as<QBox>(), (uint32_t)n, F, dKeys.as<unsigned long long>(), dVals.as<uint32_t>());
    size_t tempBytes = 0;
    GB_HIP(rocprim::radix_sort_pairs(nullptr, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(),
                                     (size_t)n, 0u, 63u, stream));
    size_t scanBytes = 0;
    GB_HIP(rocprim::exclusive_scan(nullptr, scanBytes, dKeep.as<uint32_t>(), dNewIdx.as<uint32_t>(), 0u, (size_t)nInner, rocprim::plus<uint32_t>(), stream));
    if (!dTemp.alloc(tempBytes > scanBytes ? tempBytes : scanBytes)) return "GPU BVH build: out of device memory";
    GB_HIP(rocprim::radix_sort_pairs(dTemp.p, tempBytes, dKeys.as<unsigned long long>(), dKeys2.as<unsigned long long>(), dVals.as<uint32_t>(), dVals2.as<uint32_t>(),
                                     (size_t)n, 0u, 63u, stream));
    GB_HIP(hipStreamSynchronize(stream));
    T.sortMs = (float)(now_ms() - t0); t0 = now_ms();
    const unsigned long long* keys = dKeys2.as<unsigned long long>();
    const uint32_t* sortedTri = dVals2.as<uint32_t>();

    GB_HIP(hipMemsetAsync(dArrived.p, 0, (size_t)nInner * 4, stream));
    GB_HIP(hipMemsetAsync(dParI.p, 0xFF, (size_t)nInner * 4, stream));
    GB_HIP(hipMemsetAsync(dParL.p, 0xFF, nt * 4, stream));
    hipLaunchKernelGGL(k_radix_tree, dim3(blocksI), dim3(kB), 0, stream, keys, n, dChild.as<int2>(), dRange.as<int2>(), dParI.as<int>(), dParL.as<int>());
    hipLaunchKernelGGL(k_fit, dim3(blocksN), dim3(kB), 0, stream, n, sortedTri, dLeafBox.as<QBox>(), dChild.as<int2>(), dParI.as<int>(), dParL.as<int>(),
                       dNodeBox.as<QBox>(), dArrived.as<uint32_t>());
    hipLaunchKernelGGL(k_keep, dim3(blocksI), dim3(kB), 0, stream, nInner, dRange.as<int2>(), maxLeaf, dKeep.as<uint32_t>());
    GB_HIP(rocprim::exclusive_scan(dTemp.p, scanBytes, dKeep.as<uint32_t>(), dNewIdx.as<uint32_t>(), 0u, (size_t)nInner, rocprim::plus<uint32_t>(), stream));
    uint32_t lastIdx = 0, lastKeep = 0;
    GB_HIP(hipMemcpyAsync(&lastIdx, dNewIdx.as<uint32_t>() + (nInner - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_HIP(hipMemcpyAsync(&lastKeep, dKeep.as<uint32_t>() + (nInner - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_HIP(hipStreamSynchronize(stream));
    const uint32_t nNodes2 = lastIdx + lastKeep;
    if (nNodes2 == 0 || nNodes2 > (uint32_t)nInner) return "GPU BVH build: inconsistent node count";       // (the root covers n > maxLeaf triangles: it is kept)
    DevBuf dNodes2;
    if (!dNodes2.alloc((size_t)nNodes2 * 64)) return "GPU BVH build: out of device memory";
    hipLaunchKernelGGL(k_emit2, dim3(blocksI), dim3(kB), 0, stream, nInner, dChild.as<int2>(), dRange.as<int2>(), dKeep.as<uint32_t>(), dNewIdx.as<uint32_t>(), sortedTri,
                       dLeafBox.as<QBox>(), dNodeBox.as<QBox>(), dNodes2.as<float>());
    GB_HIP(hipStreamSynchronize(stream));
    GB_HIP(hipGetLastError());
    T.treeMs = (float)(now_ms() - t0); t0 = now_ms();

    out = BvhOut();
    out.refTri.resize(nt);
    const bool fitsLds = (size_t)nNodes2 * 4 + nt * 3 <= kLdsSceneFloat4;
    const bool wide = p.nodeWidth == 4 || (p.nodeWidth == 0 && !fitsLds);
    if (!wide) {
        out.nodes.resize((size_t)nNodes2 * 16);
        out.width = 2;
        GB_HIP(hipMemcpyAsync(out.nodes.data(), dNodes2.p, (size_t)nNodes2 * 64, hipMemcpyDeviceToHost, stream));
    } else {
        // level by level: the queue of one level is [node of the BVH2, own index]; the children of a level get consecutive
        // indices in queue order, so the array is breadth-first and the same for every run
        DevBuf dNodes4, dQa, dQb, dCnt, dOff, dErr;
        const uint32_t capWide = nNodes2;              // a 4-wide node takes the place of at least one BVH2 node
        if (!dNodes4.alloc((size_t)capWide * 64) || !dQa.alloc((size_t)capWide * 8) || !dQb.alloc((size_t)capWide * 8) || !dCnt.alloc((size_t)capWide * 4) ||
            !dOff.alloc((size_t)capWide * 4) || !dErr.alloc(16))
            return "GPU BVH build: out of device memory";
        size_t lvlScanBytes = 0;
        GB_HIP(rocprim::exclusive_scan(nullptr, lvlScanBytes, dCnt.as<uint32_t>(), dOff.as<uint32_t>(), 0u, (size_t)capWide, rocprim::plus<uint32_t>(), stream));
        DevBuf dTemp2;
        if (!dTemp2.alloc(lvlScanBytes)) return "GPU BVH build: out of device memory";
        GB_HIP(hipMemsetAsync(dErr.p, 0, 16, stream));
        const uint2 rootEntry = make_uint2(0u, 0u);
        GB_HIP(hipMemcpyAsync(dQa.p, &rootEntry, 8, hipMemcpyHostToDevice, stream));
        uint32_t nq = 1, base = 1, levels = 0;          // base: index the next level's first node gets
        uint2* q = dQa.as<uint2>(); uint2* qn = dQb.as<uint2>();
        while (nq > 0) {
            if (++levels > 40) return "GPU BVH build: 4-wide tree is deeper than the traversal stack allows (use a host builder)";
            const int blocksQ = (int)((nq + kB - 1) / kB);
            hipLaunchKernelGGL(k_wide_count, dim3(blocksQ), dim3(kB), 0, stream, dNodes2.as<float>(), nNodes2, q, nq, dCnt.as<uint32_t>());
            GB_HIP(rocprim::exclusive_scan(dTemp2.p, lvlScanBytes, dCnt.as<uint32_t>(), dOff.as<uint32_t>(), 0u, (size_t)nq, rocprim::plus<uint32_t>(), stream));
            uint32_t lo = 0, lc = 0;
            GB_HIP(hipMemcpyAsync(&lo, dOff.as<uint32_t>() + (nq - 1), 4, hipMemcpyDeviceToHost, stream));
            GB_HIP(hipMemcpyAsync(&lc, dCnt.as<uint32_t>() + (nq - 1), 4, hipMemcpyDeviceToHost, stream));
            GB_HIP(hipStreamSynchronize(stream));
            const uint32_t nNext = lo + lc;
            if ((uint64_t)base + nNext > capWide) return "GPU BVH build: inconsistent 4-wide node count";
            hipLaunchKernelGGL(k_wide_emit, dim3(blocksQ), dim3(kB), 0, stream, dNodes2.as<float>(), nNodes2, q, nq, dOff.as<uint32_t>(), base, capWide,
                               dNodes4.as<float>(), qn, dErr.as<uint32_t>());
            base += nNext; nq = nNext;
            uint2* tq = q; q = qn; qn = tq;
        }
        uint32_t err = 0;
        GB_HIP(hipMemcpyAsync(&err, dErr.p, 4, hipMemcpyDeviceToHost, stream));
        GB_HIP(hipStreamSynchronize(stream));
        GB_HIP(hipGetLastError());
        if (err & 2u) return "GPU BVH build: 4-wide node index out of range";
        if (err & 1u) return "scene extent too large for quantised 4-wide nodes (use nodeWidth 2)";
        out.nodes.resize((size_t)base * 16);
        out.width = 4;
        GB_HIP(hipMemcpyAsync(out.nodes.data(), dNodes4.p, (size_t)base * 64, hipMemcpyDeviceToHost, stream));
    }
    T.collapseMs = (float)(now_ms() - t0); t0 = now_ms();
    GB_HIP(hipMemcpyAsync(out.refTri.data(), sortedTri, nt * 4, hipMemcpyDeviceToHost, stream));
    GB_HIP(hipStreamSynchronize(stream));
    T.downloadMs = (float)(now_ms() - t0);
    // depth, stack need and every index are established by validate_bvh on the host before the tree is used (api.hip)
    out.duplicates = 0;
    if (times) *times = T;
    return "";
}

}  // namespace mipt
