// bvh.h — host BVH builder interface (see bvh.cpp): binned-SAH binary tree, collapsed to the 4-wide tree the kernels read.
#pragma once
#include "lm_layout.h"
#include <vector>

struct LmBvh {
    std::vector<LmNode> nodes;          // node 0 is the root and always an inner node
    std::vector<LmNode4> nodes4;        // the tree collapsed to 4-wide nodes with 16-bit boxes (what the kernels read); node 0 = root
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    std::vector<uint32_t> order;        // BVH triangle slot -> input triangle index
    std::vector<LmWoop> woop;           // per slot, plus one all-zero sentinel packet at index order.size()
    uint32_t maxDepth = 0;              // of the binary tree
    uint32_t maxStack = 0;              // worst-case traversal stack occupancy of the 4-wide tree
    float pad = 0.f;
    std::vector<uint32_t> levelNodes;   // 4-wide node indices grouped by depth, deepest level first ...
    std::vector<uint32_t> levelStart;   // ... level l = levelNodes[levelStart[l] .. levelStart[l + 1])   (GPU refit order)
};
// tris: 9 floats per triangle (world space)
void lm_build_bvh(const float* tris, uint32_t nTris, LmBvh* out);
