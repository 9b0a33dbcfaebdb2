// bvh_builder.h -- host BVH builders that emit the GPU traversal layout directly.
//
// Layout (also documented in include/mipt.h):
//   node  = 16 floats: [c0.min xyz][c0.max xyz][c1.min xyz][c1.max xyz][ref0][ref1][0][0]
//   ref   = uint32: bit31 set -> leaf, bits 27..30 = reference count (<= 15), bits 0..26 = first reference;
//           bit31 clear -> index of an inner node.  Node 0 is the root and is always an inner node.
//   refTri[i] = global triangle id of reference i (a triangle may be referenced by several leaves: spatial splits).
//
// Wide layout (BvhParams::nodeWidth = 4, "Q4"): the BVH2 is collapsed to 4 children per node and the child boxes are
// quantised to 8 bits per plane relative to the node's own box -- still 64 bytes per node, about half as many node
// visits per ray:
//   node  = [origin xyz (3 floats)][scale.z]
//           [qlo.x of children 0..3 (4 bytes)][qlo.y][qlo.z][qhi.x][qhi.y][qhi.z][scale.x][scale.y]
//           [ref0][ref1][ref2][ref3]
//   child plane = origin + q * scale, rounded outward so the quantised box contains the child's box; the scales are
//   powers of two in [2^-60, 2^40], stored as floats.  An unused child slot holds kEmptyRef (a leaf of zero
//   references) and an inverted box (qlo = 255, qhi = 0 on every axis), which the slab test rejects by itself.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mipt {

struct TriPos { float v[3][3]; };   // three vertex positions

struct BvhParams {
    uint32_t builder = 0;        // 0 SBVH, 1 median
    uint32_t maxLeafSize = 4;    // <= 15
    float    splitAlpha = 1.0e-5f;   // SplitBVHBuilder.hpp:67
    // SAH constants: the reference's `Platform` is undeclared.  NVIDIA's default Platform uses 1 / 1; on
    // gfx950 a Moller-Trumbore test issues ~75 VALU instructions against ~50 for a two-box node visit, and the
    // measured optimum (profiles/r1/exp_sah_tricost_leafsize.log) is 1 / 1.5 for BVH2 traversal; the C-ABI picks 3.0
    // for scenes that end up in 4-wide nodes (profiles/r1/exp_sah_q4.log).
    float    nodeCost = 1.0f;    // getNodeCost(n) = nodeCost * n
    float    triCost = 1.5f;     // getTriangleCost(n) = triCost * n
    uint32_t minLeafSize = 1;
    uint32_t nodeWidth = 0;      // 2: fp32 BVH2 nodes, 4: quantised 4-wide nodes, 0: BVH2 if the scene fits the LDS window, else 4
};

struct BvhOut {
    std::vector<float>    nodes;     // 16 per node
    std::vector<uint32_t> refTri;    // per reference
    uint32_t maxDepth = 0;           // deepest inner-node level (root = 0)
    uint32_t duplicates = 0;         // references - triangles
    uint32_t width = 2;              // node layout: 2 or 4 (see above)
    uint32_t stackNeed = 0;          // upper bound of traversal-stack entries a ray can need (incl. the sentinel)
};

constexpr uint32_t kMaxBvhDepth = 64;       // traversal stack capacity on the device
constexpr uint32_t kLdsSceneFloat4 = 192;   // float4 slots of the kernels' LDS scene window (kernels.hip kLdsSceneVec)
constexpr uint32_t kMaxWideStack = 96;      // traversal stack capacity of the 4-wide kernels
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kEmptyRef = 0x80000000u; // 4-wide nodes: "no child" (a leaf of zero references)
inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) { return kLeafFlag | (count << 27) | first; }

uint64_t bvh_build_constants_hash();     // the builder's compiled-in constants (bins, depth caps, LDS window, node cost ...)

// returns empty string on success, otherwise an error message
std::string build_bvh(const std::vector<TriPos>& tris, const BvhParams& p, BvhOut& out);

// Structural check of a flat BVH that did not come straight from build_bvh (a cache file): every index in range, the
// nodes form one tree rooted at node 0 (each node reached exactly once), depth below the device stack, every triangle
// referenced.  Sets out.maxDepth / out.duplicates.  Returns empty string when the BVH is safe to traverse.
std::string validate_bvh(BvhOut& out, uint32_t nTris);     // out.width selects the layout

// Re-fits the boxes of an existing tree to moved triangles (same count; refTri and the tree's shape stay, so do
// maxDepth and stackNeed): leaf boxes = bounds of the referenced triangles, inner boxes = union of the children,
// 4-wide nodes re-quantised.  Spatial-split leaves lose their clipping (boxes stay conservative).  On error `out` is
// unchanged.
std::string refit_bvh(BvhOut& out, const std::vector<TriPos>& tris);

}  // namespace mipt
