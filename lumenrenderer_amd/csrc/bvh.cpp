// bvh.cpp — host-side BVH2 builder (binned SAH) and Woop triangle packets.
//
// Replaces the reference's two opaque calls OptixWrapper::BuildGeometryAccelerationStructure /
// BuildInstanceAccelerationStructure (LumenPT/src/Framework/OptixWrapper.cpp:46-131): instance transforms are
// baked, the whole scene becomes ONE BVH2 over world-space triangles (the reference rebuilds its instance AS on
// every transform change anyway, PTScene.cpp:145-153).  Node boxes are padded by 2^-15 * (largest |coordinate|)
// so that box tests are conservative with respect to the fp32 Woop triangle test (DESIGN.md "Traversal").
#include "bvh.h"
#include "lm_woop.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void grow(const float* p) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box& b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    float area() const { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx); }
};

struct Builder {
    const float* tris;                 // 9 floats per triangle
    std::vector<Box> tbox;
    std::vector<float> cen;            // 3 per triangle
    std::vector<uint32_t> ids;
    LmBvh* out;
    float pad;
    uint32_t maxDepth = 0;

    struct Ref { int ref; Box box; };

    static int ilog2ceil(uint32_t v) { int r = 0; while ((1u << r) < v) r++; return r; }

    Ref makeLeaf(uint32_t first, uint32_t count, const Box& box)
    {
        const uint32_t start = (uint32_t)out->order.size();
        for (uint32_t i = 0; i < count; i++) out->order.push_back(ids[first + i]);
        Ref r; r.ref = ~(int)((start << 3) | (count - 1u)); r.box = box;
        return r;
    }

    Ref build(uint32_t first, uint32_t count, uint32_t depth)
    {
        maxDepth = std::max(maxDepth, depth);
        Box box; box.reset();
        Box cbox; cbox.reset();
        for (uint32_t i = first; i < first + count; i++) { box.grow(tbox[ids[i]]); cbox.grow(&cen[3 * ids[i]]); }
        if (count <= 2) return makeLeaf(first, count, box);
        // depth guard: from here a median split is guaranteed to finish within the traversal stack
        const bool forceMedian = (int)depth + ilog2ceil(count) + 2 >= LM_BVH2_MAX_DEPTH - 2;
        uint32_t mid = 0;
        bool split = false;
        if (!forceMedian) {
            const int NB = 16;
            float bestCost = INFINITY; int bestAxis = -1, bestBin = -1;
            for (int axis = 0; axis < 3; axis++) {
                const float lo = cbox.lo[axis], ext = cbox.hi[axis] - lo;
                if (!(ext > 0.f)) continue;
                Box bb[NB]; uint32_t bc[NB];
                for (int b = 0; b < NB; b++) { bb[b].reset(); bc[b] = 0; }
                const float scale = (float)NB / ext;
                for (uint32_t i = first; i < first + count; i++) {
                    int b = (int)((cen[3 * ids[i] + axis] - lo) * scale);
                    b = std::min(NB - 1, std::max(0, b));
                    bb[b].grow(tbox[ids[i]]); bc[b]++;
                }
                float rightArea[NB]; uint32_t rightCount[NB];
                Box acc; acc.reset(); uint32_t cnt = 0;
                for (int b = NB - 1; b > 0; b--) { acc.grow(bb[b]); cnt += bc[b]; rightArea[b] = acc.area(); rightCount[b] = cnt; }
                acc.reset(); cnt = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bb[b]); cnt += bc[b];
                    if (cnt == 0 || rightCount[b + 1] == 0) continue;
                    const float cost = acc.area() * (float)cnt + rightArea[b + 1] * (float)rightCount[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
            const float leafCost = box.area() * (float)count;
            if (bestAxis >= 0 && (count > 4 || bestCost + box.area() * 1.0f < leafCost)) {
                const float lo = cbox.lo[bestAxis], ext = cbox.hi[bestAxis] - lo;
                const float scale = 16.f / ext;
                auto it = std::partition(ids.begin() + first, ids.begin() + first + count, [&](uint32_t t) {
                    int b = (int)((cen[3 * t + bestAxis] - lo) * scale);
                    b = std::min(15, std::max(0, b));
                    return b <= bestBin;
                });
                mid = (uint32_t)(it - ids.begin());
                split = mid > first && mid < first + count;
            } else if (count <= 4) {
                return makeLeaf(first, count, box);
            }
        }
        if (!split) {
            if (count <= LM_MAX_LEAF && !forceMedian) return makeLeaf(first, count, box);
            if (count <= 4) return makeLeaf(first, count, box);
            int axis = 0;
            for (int k = 1; k < 3; k++) if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
            mid = first + count / 2;
            std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count, [&](uint32_t a, uint32_t b) {
                const float ka = cen[3 * a + axis], kb = cen[3 * b + axis];
                return ka < kb || (ka == kb && a < b);
            });
        }
        const int self = (int)out->nodes.size();
        out->nodes.emplace_back();
        const Ref l = build(first, mid - first, depth + 1);
        const Ref r = build(mid, first + count - mid, depth + 1);
        setNode(self, l, r);
        Ref me; me.ref = self; me.box = box;
        return me;
    }

    void setNode(int idx, const Ref& l, const Ref& r)
    {
        LmNode& n = out->nodes[idx];
        auto P = [&](float v, float s) { return v + s * pad; };
        n.n0 = make_float4(P(l.box.lo[0], -1), P(l.box.hi[0], 1), P(l.box.lo[1], -1), P(l.box.hi[1], 1));
        n.n1 = make_float4(P(r.box.lo[0], -1), P(r.box.hi[0], 1), P(r.box.lo[1], -1), P(r.box.hi[1], 1));
        n.n2 = make_float4(P(l.box.lo[2], -1), P(l.box.hi[2], 1), P(r.box.lo[2], -1), P(r.box.hi[2], 1));
        n.ref = make_int4(l.ref, r.ref, 0, 0);
    }
};

}  // namespace

void lm_build_bvh(const float* tris, uint32_t nTris, LmBvh* out)
{
    out->nodes.clear(); out->order.clear(); out->woop.clear();
    Builder b;
    b.tris = tris; b.out = out;
    b.tbox.resize(nTris); b.cen.resize(3 * (size_t)nTris); b.ids.resize(nTris);
    float maxAbs = 0.f;
    for (uint32_t t = 0; t < nTris; t++) {
        b.tbox[t].reset();
        for (int v = 0; v < 3; v++) {
            const float* p = tris + 9 * (size_t)t + 3 * v;
            b.tbox[t].grow(p);
            for (int k = 0; k < 3; k++) maxAbs = std::max(maxAbs, std::fabs(p[k]));
        }
        for (int k = 0; k < 3; k++) b.cen[3 * (size_t)t + k] = 0.5f * (b.tbox[t].lo[k] + b.tbox[t].hi[k]);
        b.ids[t] = t;
    }
    b.pad = maxAbs * (1.0f / 32768.0f);
    out->pad = b.pad;
    // absent child: marked by NaN boxes here; the quantised tree points it at the sentinel (never hit) triangle packet
    Builder::Ref empty; empty.ref = ~0;
    for (int k = 0; k < 3; k++) { empty.box.lo[k] = NAN; empty.box.hi[k] = NAN; }
    if (nTris == 0) {
        out->nodes.emplace_back();
        b.setNode(0, empty, empty);
    } else {
        out->nodes.reserve(nTris);
        out->order.reserve(nTris);
        Builder::Ref root = b.build(0, nTris, 0);
        if (root.ref < 0) {                                    // whole scene fits one leaf: node 0 must still be an inner node
            out->nodes.emplace_back();
            b.setNode(0, root, empty);
        }
    }
    out->maxDepth = b.maxDepth + 1;
    const size_t nSlots = out->order.size();
    out->woop.resize(nSlots + 1);
    for (size_t s = 0; s < nSlots; s++) out->woop[s] = lm_make_woop(tris + 9 * (size_t)out->order[s]);
    memset(&out->woop[nSlots], 0, sizeof(LmWoop));               // sentinel packet: t = -0/0 = NaN, never a hit

    // ---- 16-bit quantisation relative to the (padded) scene box, rounded outward by one extra step
    float smin[3] = {INFINITY, INFINITY, INFINITY}, smax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = 0; t < nTris; t++) for (int k = 0; k < 3; k++) { smin[k] = std::min(smin[k], b.tbox[t].lo[k]); smax[k] = std::max(smax[k], b.tbox[t].hi[k]); }
    for (int k = 0; k < 3; k++) {
        if (!(smin[k] <= smax[k])) { smin[k] = 0.f; smax[k] = 1.f; }
        const float m = 4.f * b.pad + 1e-6f * std::max(std::fabs(smin[k]), std::fabs(smax[k])) + 1e-30f;
        smin[k] -= m; smax[k] += m;
        out->qmin[k] = smin[k];
        out->qstep[k] = (smax[k] - smin[k]) / 65535.0f;
    }
    auto quant = [&](float lo, float hi, int k, uint32_t& packed) {
        const double inv = 1.0 / (double)out->qstep[k];
        long long ql = (long long)std::floor(((double)lo - (double)out->qmin[k]) * inv) - 1;
        long long qh = (long long)std::ceil(((double)hi - (double)out->qmin[k]) * inv) + 1;
        ql = std::max(0LL, std::min(65535LL, ql)); qh = std::max(0LL, std::min(65535LL, qh));
        packed = (uint32_t)ql | ((uint32_t)qh << 16);
    };
    // ---- collapse to 4-wide nodes: starting from a binary node's two children, repeatedly replace the inner child of
    // largest surface area by its own two children until there are four (or only leaves are left)
    struct Child { int ref; float b[6]; };      // padded box: lo.x hi.x lo.y hi.y lo.z hi.z
    auto childrenOf = [&](int node, Child* c) {
        const LmNode& n = out->nodes[node];
        const float c0[6] = {n.n0.x, n.n0.y, n.n0.z, n.n0.w, n.n2.x, n.n2.y}, c1[6] = {n.n1.x, n.n1.y, n.n1.z, n.n1.w, n.n2.z, n.n2.w};
        int k = 0;
        if (c0[0] == c0[0]) { c[k].ref = n.ref.x; memcpy(c[k].b, c0, sizeof c0); k++; }
        if (c1[0] == c1[0]) { c[k].ref = n.ref.y; memcpy(c[k].b, c1, sizeof c1); k++; }
        return k;
    };
    auto areaOf = [](const Child& c) { const float dx = c.b[1] - c.b[0], dy = c.b[3] - c.b[2], dz = c.b[5] - c.b[4]; return dx * dy + dy * dz + dz * dx; };
    out->nodes4.clear();
    out->nodes4.reserve(out->nodes.size() / 2 + 1);
    out->maxStack = 1;
    struct Work { int node2; int node4; uint32_t stack; uint32_t depth; };
    std::vector<uint32_t> depthOf;
    std::vector<Work> work;
    out->nodes4.emplace_back();
    work.push_back({0, 0, 0, 0});
    depthOf.push_back(0);
    while (!work.empty()) {
        const Work w = work.back(); work.pop_back();
        Child c[4];
        int n = childrenOf(w.node2, c);
        while (n < 4) {
            int best = -1; float bestArea = -1.f;
            for (int i = 0; i < n; i++) if (c[i].ref >= 0) { const float a = areaOf(c[i]); if (a > bestArea) { bestArea = a; best = i; } }
            if (best < 0) break;
            Child g[2];
            const int m = childrenOf(c[best].ref, g);
            if (n - 1 + m > 4) break;
            c[best] = g[0];
            if (m > 1) c[n++] = g[1];
        }
        const uint32_t stackBelow = w.stack + (uint32_t)(n > 0 ? n - 1 : 0);
        out->maxStack = std::max(out->maxStack, stackBelow + 1u);
        LmNode4 q;
        for (int i = 0; i < 4; i++) {
            if (i >= n) { q.c[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, (uint32_t)LM_REF_NONE); continue; }
            uint32_t p[3];
            for (int k = 0; k < 3; k++) quant(c[i].b[2 * k], c[i].b[2 * k + 1], k, p[k]);
            int ref = c[i].ref;
            if (ref >= 0) {
                const int id4 = (int)out->nodes4.size();
                out->nodes4.emplace_back();
                work.push_back({ref, id4, stackBelow, w.depth + 1});
                depthOf.push_back(w.depth + 1);
                ref = id4;
            }
            q.c[i] = make_uint4(p[0], p[1], p[2], (uint32_t)ref);
        }
        out->nodes4[w.node4] = q;
    }
    // nodes grouped by depth (deepest level first): the order in which a bottom-up refit visits them
    uint32_t maxD = 0;
    for (uint32_t d : depthOf) maxD = std::max(maxD, d);
    out->levelStart.assign(maxD + 2, 0);
    for (uint32_t d : depthOf) out->levelStart[maxD - d + 1]++;
    for (uint32_t l = 0; l <= maxD; l++) out->levelStart[l + 1] += out->levelStart[l];
    out->levelNodes.resize(depthOf.size());
    std::vector<uint32_t> fill(out->levelStart.begin(), out->levelStart.end() - 1);
    for (uint32_t i = 0; i < depthOf.size(); i++) out->levelNodes[fill[maxD - depthOf[i]]++] = i;
}
