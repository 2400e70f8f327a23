// Host-side check of the BVH builder the product links (lumenrenderer_amd/csrc/bvh.cpp), compiled for the CPU by
// tests/test_cpu_host.py (plain, and under ThreadSanitizer / AddressSanitizer+UBSan).  Test infrastructure only.
//   structure    : the triangle order is a permutation; the leaves of the 4-wide tree partition the slots; every quantised child
//                  box contains the triangles below it; the refit level lists are bottom-up and complete; the reported worst-case
//                  stack occupancy covers the tree and fits the kernels' stack
//   Woop packets : bit-identical to lm_make_packet (the function the GPU refit runs) of the ordered triangle, zero sentinel
//   determinism  : 1 thread and N threads produce the same bytes
//   assembly     : lm_assemble_bvh (instance-level trees for topology edits) passes the same structural checks
// usage: bvh_check <threads> <nTris>...      prints "ok <n> ..." per size, exits non-zero on the first violation
#include "bvh.h"
#include "lm_tri.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

uint32_t g_state = 1;
float rnd() { g_state ^= g_state << 13; g_state ^= g_state >> 17; g_state ^= g_state << 5; return (float)(g_state >> 8) * (1.0f / 16777216.0f); }

std::vector<float> soup(uint32_t n, uint32_t seed)
{
    g_state = seed * 2654435761u + 12345u;
    std::vector<float> t(9 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        const float cx = rnd() * 40.f - 20.f, cy = rnd() * 10.f, cz = rnd() * 60.f - 30.f;
        const float s = (i % 97 == 0) ? 8.f : 0.05f + rnd() * 0.6f;                 // a few large triangles among many small ones
        for (int v = 0; v < 3; v++) { t[9 * i + 3 * v] = cx + (rnd() - 0.5f) * s; t[9 * i + 3 * v + 1] = cy + (rnd() - 0.5f) * s; t[9 * i + 3 * v + 2] = cz + (rnd() - 0.5f) * s; }
        if (i % 31 == 7) for (int k = 0; k < 3; k++) t[9 * i + 6 + k] = t[9 * i + 3 + k];               // degenerate: two equal vertices
        if (i % 53 == 11 && i > 0) memcpy(&t[9 * i], &t[9 * (i - 1)], 9 * sizeof(float));                 // exact duplicate of the previous triangle
        if (i % 41 == 3) { for (int v = 0; v < 3; v++) t[9 * i + 3 * v + 1] = 2.5f; }                     // axis-aligned (flat box)
    }
    return t;
}

int fail(const char* what, long a = 0, long b = 0) { fprintf(stderr, "bvh_check: %s (%ld, %ld)\n", what, a, b); return 1; }

struct Walk {
    const LmBvh& b; const float* tris; bool boxes = true;
    std::vector<uint8_t> slotSeen, nodeSeen;
    std::vector<int> nodeDepth;
    int err = 0;
    Walk(const LmBvh& bb, const float* t) : b(bb), tris(t), slotSeen(bb.order.size(), 0), nodeSeen(bb.nodesW.size(), 0), nodeDepth(bb.nodesW.size(), -1) {}
    // returns the exact box of the subtree in lo/hi and the worst-case stack occupancy below this reference
    uint32_t visit(int ref, int depth, double lo[3], double hi[3])
    {
        for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
        if (ref < 0) {
            const uint32_t leaf = (uint32_t)(~ref), first = leaf >> 3, cnt = (leaf & 7u) + 1u;
            if (first + cnt > b.order.size()) { err = fail("leaf range outside the slots", first, cnt); return 0; }
            for (uint32_t s = first; s < first + cnt; s++) {
                if (slotSeen[s]++) { err = fail("slot in two leaves", s); return 0; }
                const float* t = tris + 9 * (size_t)b.order[s];
                for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)t[3 * v + k]); hi[k] = std::max(hi[k], (double)t[3 * v + k]); }
            }
            return 0;
        }
        if ((size_t)ref >= b.nodesW.size()) { err = fail("node reference out of range", ref); return 0; }
        if (nodeSeen[ref]++) { err = fail("node reachable twice", ref); return 0; }
        nodeDepth[ref] = depth;
        uint32_t present = 0, worst = 0;
        for (int c = 0; c < LM_WIDTH; c++) present += (int)b.nodesW[ref].c[c].w != LM_REF_NONE;
        if (present == 0 && !(ref == 0 && b.order.empty())) { err = fail("inner node without children", ref); return 0; }
        for (int c = 0; c < LM_WIDTH && !err; c++) {
            const uint4 q = b.nodesW[ref].c[c];
            if ((int)q.w == LM_REF_NONE) continue;
            double clo[3], chi[3];
            const uint32_t below = visit((int)q.w, depth + 1, clo, chi);
            worst = std::max(worst, present - 1u + below);
            const uint32_t qq[3] = {q.x, q.y, q.z};
            for (int k = 0; k < 3 && !err; k++) {
                const double blo = (double)b.qmin[k] + (double)(qq[k] & 0xffffu) * (double)b.qstep[k];
                const double bhi = (double)b.qmin[k] + (double)(qq[k] >> 16) * (double)b.qstep[k];
                // (the kernels evaluate the box in fp32: allow their rounding, which the builder's outward padding covers)
                const double eps = 4.0 * (double)b.pad;
                if (boxes && clo[k] <= chi[k] && (blo > clo[k] + eps || bhi < chi[k] - eps)) err = fail("child box does not contain its triangles", ref, c);
                lo[k] = std::min(lo[k], clo[k]); hi[k] = std::max(hi[k], chi[k]);
            }
        }
        return worst;
    }
};

int check(const LmBvh& b, const float* tris, uint32_t n, bool built = true)
{
    if (b.order.size() != n) return fail("order size", (long)b.order.size(), n);
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t s = 0; s < n; s++) { if (b.order[s] >= n || seen[b.order[s]]++) return fail("order is not a permutation", s); }
    if (b.nodesW.empty()) return fail("no root node");
    if (b.packets.size() != (size_t)n + 1) return fail("packets packet count", (long)b.packets.size());
    for (uint32_t s = 0; built && s < n; s++) {
        const LmTriPacket w = lm_make_packet(tris + 9 * (size_t)b.order[s]);
        if (memcmp(&w, &b.packets[s], sizeof w) != 0) return fail("packets packet differs from lm_make_packet", s);
    }
    { LmTriPacket z; memset(&z, 0, sizeof z); if (memcmp(&z, &b.packets[n], sizeof z) != 0) return fail("sentinel packet is not zero"); }
    Walk w(b, tris);
    w.boxes = built;
    double lo[3], hi[3];
    const uint32_t worst = w.visit(0, 0, lo, hi);
    if (w.err) return 1;
    for (uint32_t s = 0; s < n; s++) if (!w.slotSeen[s]) return fail("slot in no leaf", s);
    for (size_t i = 0; i < b.nodesW.size(); i++) if (!w.nodeSeen[i]) return fail("unreachable node", (long)i);
    if (b.maxStack < worst) return fail("maxStack below the worst case of the tree", b.maxStack, worst);
    if (b.maxStack > LM_STACK_DEPTH) return fail("tree needs more than LM_STACK_DEPTH", b.maxStack);
    // refit order: every node exactly once, the children of a node in an earlier (deeper) level
    if (b.levelNodes.size() != b.nodesW.size() || b.levelStart.empty() || b.levelStart.back() != b.levelNodes.size()) return fail("level lists incomplete");
    std::vector<int> levelOf(b.nodesW.size(), -1);
    for (size_t l = 0; l + 1 < b.levelStart.size(); l++)
        for (uint32_t i = b.levelStart[l]; i < b.levelStart[l + 1]; i++) {
            const uint32_t nd = b.levelNodes[i];
            if (nd >= b.nodesW.size() || levelOf[nd] != -1) return fail("level list entry", nd);
            levelOf[nd] = (int)l;
        }
    for (size_t nd = 0; nd < b.nodesW.size(); nd++)
        for (int c = 0; c < LM_WIDTH; c++) {
            const int ref = (int)b.nodesW[nd].c[c].w;
            if (ref >= 0 && ref != LM_REF_NONE && !(levelOf[ref] < levelOf[nd])) return fail("child not refitted before its parent", (long)nd, ref);
        }
    return 0;
}

bool same(const LmBvh& a, const LmBvh& b)
{
    auto eq = [](const auto& x, const auto& y) { return x.size() == y.size() && (x.empty() || memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0); };
    return eq(a.nodesW, b.nodesW) && eq(a.order, b.order) && eq(a.packets, b.packets) && eq(a.levelNodes, b.levelNodes) && eq(a.levelStart, b.levelStart) &&
           memcmp(a.qmin, b.qmin, sizeof a.qmin) == 0 && memcmp(a.qstep, b.qstep, sizeof a.qstep) == 0 && a.pad == b.pad && a.maxStack == b.maxStack && a.maxDepth == b.maxDepth;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: bvh_check <threads> <nTris>...\n"); return 2; }
    const std::string threads = argv[1];
    for (int a = 2; a < argc; a++) {
        const uint32_t n = (uint32_t)atoi(argv[a]);
        const std::vector<float> t = soup(n, (uint32_t)a);
        LmBvh one, many;
        setenv("LUMEN_MI_BUILD_THREADS", "1", 1);
        lm_build_bvh(t.data(), n, &one);
        setenv("LUMEN_MI_BUILD_THREADS", threads.c_str(), 1);
        lm_build_bvh(t.data(), n, &many);
        if (check(one, t.data(), n) || check(many, t.data(), n)) return 1;
        if (!same(one, many)) return fail("the build depends on the thread count", n);
        printf("ok %u triangles: %zu nodes, depth %u, stack %u\n", n, many.nodesW.size(), many.maxDepth, many.maxStack);
    }
    // instance-level assembly (lm_assemble_bvh): per-mesh trees + a top tree; topology only (boxes / packets come from the GPU refit)
    for (uint32_t nInst : {1u, 2u, 3u, 5u, 9u, 40u}) {
        const uint32_t sizes[3] = {1u, 37u, 2500u};
        std::vector<float> meshTris[3]; LmBvh meshBvh[3];
        for (int m = 0; m < 3; m++) { meshTris[m] = soup(sizes[m], 100u + (uint32_t)m); lm_build_bvh(meshTris[m].data(), sizes[m], &meshBvh[m]); }
        std::vector<LmInstanceRef> inst(nInst);
        std::vector<float> world;
        for (uint32_t i = 0; i < nInst; i++) {
            const int m = (int)(i % 3u);
            const float off[3] = {50.f * (float)(i % 4u), 7.f * (float)(i / 4u), -30.f * (float)(i % 3u)};
            inst[i].mesh = &meshBvh[m]; inst[i].triBase = (uint32_t)(world.size() / 9);
            for (int k = 0; k < 3; k++) { inst[i].box[k] = 1e30f; inst[i].box[3 + k] = -1e30f; }
            for (size_t f = 0; f < meshTris[m].size(); f++) {
                const float v = meshTris[m][f] + off[f % 3];
                world.push_back(v);
                inst[i].box[f % 3] = std::min(inst[i].box[f % 3], v); inst[i].box[3 + f % 3] = std::max(inst[i].box[3 + f % 3], v);
            }
        }
        LmBvh scene;
        lm_assemble_bvh(inst.data(), nInst, &scene);
        const uint32_t n = (uint32_t)(world.size() / 9);
        if (check(scene, world.data(), n, false)) return 1;
        { bool any = false; for (int c = 0; c < LM_WIDTH; c++) any |= (int)scene.nodesW[0].c[c].w != LM_REF_NONE;     // (8-wide: children sit in octant slots, slot 0 may be empty)
          if (scene.nodesW.size() > 0 && !any) return fail("assembled root has no child"); }
        printf("ok assembly of %u instances: %u triangles, %zu nodes, depth %u, stack %u\n", nInst, n, scene.nodesW.size(), scene.maxDepth, scene.maxStack);
    }
    return 0;
}
