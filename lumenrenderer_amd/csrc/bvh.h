// bvh.h — host BVH2 builder interface (see bvh.cpp).
#pragma once
#include "lm_layout.h"
#include <vector>

struct LmBvh {
    std::vector<LmNode> nodes;          // node 0 is the root and always an inner node
    std::vector<LmNodeQ> qnodes;        // the same tree with 16-bit boxes (what the kernels read)
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    std::vector<uint32_t> order;        // BVH triangle slot -> input triangle index
    std::vector<LmWoop> woop;           // per slot, plus one all-zero sentinel packet at index order.size()
    uint32_t maxDepth = 0;
    float pad = 0.f;
};
// tris: 9 floats per triangle (world space)
void lm_build_bvh(const float* tris, uint32_t nTris, LmBvh* out);
LmWoop lm_make_woop(const float* tri9);
