// gpu_bvh.h -- BVH build on the GPU ("parallel GPU/CPU BVH build"; the reference builds on the host only,
// CUDATracer/Types.hpp:312-413).  A linear BVH: 63-bit Morton codes of the triangle centroids, one device-wide radix sort,
// the binary radix tree of Karras 2012 built with one thread per inner node, boxes bottom-up, subtrees of at most
// maxLeafSize triangles folded into leaves, and -- for the 4-wide layout -- the same collapse rule and the same quantisation
// routine as the host builder (bvh_quantise.h), level by level.  Output is the flat layout of bvh_builder.h, so everything
// downstream (validation, upload, refit, cache, the oracle's traversal of the same bytes) is unchanged.
// The tree is a few times faster to build and slower to trace than the SBVH (no SAH, no spatial splits): it is for scenes
// that are rebuilt often or are too large to wait for.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "bvh_builder.h"

namespace mipt {

struct GpuBuildTimes { float uploadMs = 0, sortMs = 0, treeMs = 0, collapseMs = 0, downloadMs = 0; };

constexpr uint32_t kGpuBuildMinTris = 64;      // below this the host builders are instant and the LDS layout wants their tree

// returns empty string on success, otherwise an error message.  The device of `stream` must be current.
std::string build_bvh_gpu(const std::vector<TriPos>& tris, const BvhParams& p, BvhOut& out, hipStream_t stream, GpuBuildTimes* times);

}  // namespace mipt
