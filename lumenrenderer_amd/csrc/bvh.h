// bvh.h — host BVH builder interface (see bvh.cpp): binned-SAH binary tree, collapsed to the 4-wide tree the kernels read.
#pragma once
#include "lm_layout.h"
#include <vector>

struct LmBvh {
    std::vector<LmNode> nodes;          // node 0 is the root and always an inner node
    std::vector<LmNodeW> nodesW;        // the tree collapsed to 4-wide nodes with 16-bit boxes (what the kernels read); node 0 = root
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    std::vector<uint32_t> order;        // BVH triangle slot -> input triangle index
    std::vector<LmTriPacket> packets;           // per slot, plus one all-zero sentinel packet at index order.size()
    uint32_t maxDepth = 0;              // of the binary tree
    uint32_t maxStack = 0;              // worst-case traversal stack occupancy of the 4-wide tree
    float pad = 0.f;
    std::vector<uint32_t> levelNodes;   // 4-wide node indices grouped by depth, deepest level first ...
    std::vector<uint32_t> levelStart;   // ... level l = levelNodes[levelStart[l] .. levelStart[l + 1])   (GPU refit order)
};
// tris: 9 floats per triangle (world space)
void lm_build_bvh(const float* tris, uint32_t nTris, LmBvh* out);

// Instance-level assembly for topology edits (an instance added or removed): the scene tree becomes a small 4-wide top tree over the
// instances plus the cached tree of every instance's mesh (built once, in object space) copied behind it with node / slot offsets.
// Topology only: child boxes and Woop packets of the result are placeholders, the GPU refit (kernels.hip lm_k_refit_*) computes
// them from the instance transforms.  Hit records do not depend on the tree, so the image equals that of a full rebuild.
struct LmInstanceRef {
    const LmBvh* mesh;          // lm_build_bvh over the mesh's object-space triangles (at least one triangle)
    float box[6];               // world-space box of the instance, lo.xyz hi.xyz (only to shape the top tree)
    uint32_t triBase;           // global index of the instance's first triangle (the tie-break order of the hit rule)
};
void lm_assemble_bvh(const LmInstanceRef* inst, uint32_t nInst, LmBvh* out);

// Device-side build of the scene tree (bvh_gpu.hip): topology only — child references, triangle order, depth levels, stack need; boxes and packets come from the
// refit kernels.  Inputs are device pointers: the instance table, the vertex / index pools and per input triangle its (table entry, primitive-local triangle).
// Returns 0, or non-zero when the caller should use lm_build_bvh instead.
int lm_build_bvh_gpu(hipStream_t stream, const LmEntry* dEntries, const float4* dVerts, const uint32_t* dIndices, const uint2* dTriIn, uint32_t nTris, LmBvh* out);
