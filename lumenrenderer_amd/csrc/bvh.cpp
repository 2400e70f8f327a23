// bvh.cpp — host-side BVH builder: binned-SAH binary tree collapsed to the LM_WIDTH-wide tree the kernels read (8-wide: children in octant
// slots), Woop triangle packets.
//
// Replaces the reference's two opaque calls OptixWrapper::BuildGeometryAccelerationStructure /
// BuildInstanceAccelerationStructure (LumenPT/src/Framework/OptixWrapper.cpp:46-131): instance transforms are
// baked, the whole scene becomes ONE tree over world-space triangles (the reference rebuilds its instance AS on
// every transform change anyway, PTScene.cpp:145-153).  Node boxes are padded by 2^-15 * (largest |coordinate|)
// so that box tests are conservative with respect to the fp32 Woop triangle test (DESIGN.md "Traversal").
#include "bvh.h"
#include "lm_tri.h"

#include <sched.h>

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void grow(const float* p) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box& b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    float area() const { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx); }
};

// parallel-for over [0, n) in contiguous chunks (used for the few very large nodes at the top of the tree)
template <class F> void parallelChunks(size_t n, unsigned threads, F f)
{
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 65536 + 1));
    if (threads == 1) { f(0u, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++) pool.emplace_back([=] { f(t, n * t / threads, n * (t + 1) / threads); });
    for (auto& th : pool) th.join();
}

struct Task { uint32_t first, count, depth; };

// SAH constants (tunable for experiments through LUMEN_MI_BVH_TRAV_COST / LUMEN_MI_BVH_LEAF_MAX)
float g_travCost = 1.0f;       // cost of one node step relative to one triangle test
uint32_t g_leafMax = 4;        // largest leaf the SAH may choose (LM_MAX_LEAF = 8 is the format's limit)
uint32_t g_sweepBelow = 0;     // nodes of at most this many triangles are split by an exact SAH sweep over the sorted centroids of each axis instead of 16 bins
                               // (LUMEN_MI_BVH_SWEEP; 0 = binned everywhere).  Measured on C2: see profiles/r03_bvh_sweep_ab.txt

// One builder instance = one output arena (nodes + leaf order with LOCAL indices).  The top-level instance cuts subtrees of
// at most `taskThreshold` triangles into tasks; the task builders run in parallel on disjoint ranges of the shared index
// array, and the arenas are concatenated in task order afterwards (so the result does not depend on thread scheduling).
struct Builder {
    const Box* tbox = nullptr;         // per triangle
    const float* cen = nullptr;        // 3 per triangle
    uint32_t* ids = nullptr;           // shared permutation; a builder only touches its own range
    std::vector<LmNode> nodes;
    std::vector<uint32_t> order;
    std::vector<Task>* tasks = nullptr;
    uint32_t taskThreshold = 0;
    unsigned threads = 1;
    float pad = 0.f;
    uint32_t maxDepth = 0;

    struct Ref { int ref; Box box; };
    static constexpr int TASK_REF = 0x40000000;        // placeholder reference: TASK_REF + task index

    static int ilog2ceil(uint32_t v) { int r = 0; while ((1u << r) < v) r++; return r; }

    Ref makeLeaf(uint32_t first, uint32_t count, const Box& box)
    {
        const uint32_t start = (uint32_t)order.size();
        for (uint32_t i = 0; i < count; i++) order.push_back(ids[first + i]);
        Ref r; r.ref = ~(int)((start << 3) | (count - 1u)); r.box = box;
        return r;
    }

    Ref build(uint32_t first, uint32_t count, uint32_t depth)
    {
        maxDepth = std::max(maxDepth, depth);
        const bool big = tasks != nullptr && threads > 1 && count >= 65536u;
        Box box; box.reset();
        Box cbox; cbox.reset();
        if (big) {
            std::vector<Box> pb(threads), pc(threads);
            for (auto& x : pb) x.reset();
            for (auto& x : pc) x.reset();
            parallelChunks(count, threads, [&](unsigned t, size_t lo, size_t hi) {
                Box b0, c0; b0.reset(); c0.reset();
                for (size_t i = first + lo; i < first + hi; i++) { b0.grow(tbox[ids[i]]); c0.grow(&cen[3 * (size_t)ids[i]]); }
                pb[t] = b0; pc[t] = c0;
            });
            for (unsigned t = 0; t < threads; t++) { if (pb[t].lo[0] <= pb[t].hi[0]) box.grow(pb[t]); if (pc[t].lo[0] <= pc[t].hi[0]) cbox.grow(pc[t]); }
        } else {
            for (uint32_t i = first; i < first + count; i++) { box.grow(tbox[ids[i]]); cbox.grow(&cen[3 * (size_t)ids[i]]); }
        }
        if (tasks && depth > 0 && count <= taskThreshold && count > 2) {
            Ref r; r.ref = TASK_REF + (int)tasks->size(); r.box = box;
            tasks->push_back(Task{first, count, depth});
            return r;
        }
        if (count <= 2) return makeLeaf(first, count, box);
        // depth guard: from here a median split is guaranteed to finish within the traversal stack
        const bool forceMedian = (int)depth + ilog2ceil(count) + 2 >= LM_BVH2_MAX_DEPTH - 2;
        uint32_t mid = 0;
        bool split = false;
        if (!forceMedian && count <= g_sweepBelow) {
            // exact sweep: every position of the centroid order of every axis is a candidate (same cost function as the bins below)
            float bestCost = INFINITY; int bestAxis = -1; uint32_t bestK = 0;
            std::vector<float> rightArea(count);
            for (int axis = 0; axis < 3; axis++) {
                if (!(cbox.hi[axis] > cbox.lo[axis])) continue;
                std::sort(ids + first, ids + first + count, [&](uint32_t a, uint32_t b) {
                    const float ka = cen[3 * (size_t)a + axis], kb = cen[3 * (size_t)b + axis];
                    return ka < kb || (ka == kb && a < b);
                });
                Box acc; acc.reset();
                for (uint32_t k = count - 1; k > 0; k--) { acc.grow(tbox[ids[first + k]]); rightArea[k] = acc.area(); }
                acc.reset();
                for (uint32_t k = 0; k + 1 < count; k++) {
                    acc.grow(tbox[ids[first + k]]);
                    const float cost = acc.area() * (float)(k + 1) + rightArea[k + 1] * (float)(count - k - 1);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestK = k + 1; }
                }
            }
            const float leafCost = box.area() * (float)count;
            if (bestAxis >= 0 && (count > g_leafMax || bestCost + box.area() * g_travCost < leafCost)) {
                if (bestAxis != 2) std::sort(ids + first, ids + first + count, [&](uint32_t a, uint32_t b) {      // (the ids are in the order of the last axis swept)
                    const float ka = cen[3 * (size_t)a + bestAxis], kb = cen[3 * (size_t)b + bestAxis];
                    return ka < kb || (ka == kb && a < b);
                });
                else if (!(cbox.hi[2] > cbox.lo[2])) std::sort(ids + first, ids + first + count, [&](uint32_t a, uint32_t b) {
                    const float ka = cen[3 * (size_t)a + bestAxis], kb = cen[3 * (size_t)b + bestAxis];
                    return ka < kb || (ka == kb && a < b);
                });
                mid = first + bestK;
                split = true;
            } else if (count <= g_leafMax) {
                return makeLeaf(first, count, box);
            }
        } else if (!forceMedian) {
            const int NB = 16;
            float bestCost = INFINITY; int bestAxis = -1, bestBin = -1;
            // bins of all three axes in one pass (per thread for big nodes, merged in thread order)
            struct Bins { Box bb[3][NB]; uint32_t bc[3][NB]; };
            auto clearBins = [&](Bins& B) { for (int a = 0; a < 3; a++) for (int k = 0; k < NB; k++) { B.bb[a][k].reset(); B.bc[a][k] = 0; } };
            float lo3[3], scale3[3]; bool use[3];
            for (int axis = 0; axis < 3; axis++) { lo3[axis] = cbox.lo[axis]; const float ext = cbox.hi[axis] - lo3[axis]; use[axis] = ext > 0.f; scale3[axis] = use[axis] ? (float)NB / ext : 0.f; }
            auto binRange = [&](Bins& B, size_t a0, size_t a1) {
                for (size_t i = a0; i < a1; i++) {
                    const uint32_t t = ids[i];
                    for (int axis = 0; axis < 3; axis++) {
                        if (!use[axis]) continue;
                        int k = (int)((cen[3 * (size_t)t + axis] - lo3[axis]) * scale3[axis]);
                        k = std::min(NB - 1, std::max(0, k));
                        B.bb[axis][k].grow(tbox[t]); B.bc[axis][k]++;
                    }
                }
            };
            Bins all; clearBins(all);
            if (big) {
                std::vector<Bins> part(threads);
                for (auto& x : part) clearBins(x);
                parallelChunks(count, threads, [&](unsigned t, size_t lo, size_t hi) { binRange(part[t], first + lo, first + hi); });
                for (unsigned t = 0; t < threads; t++) for (int a = 0; a < 3; a++) for (int k = 0; k < NB; k++) if (part[t].bc[a][k]) { all.bb[a][k].grow(part[t].bb[a][k]); all.bc[a][k] += part[t].bc[a][k]; }
            } else {
                binRange(all, first, (size_t)first + count);
            }
            for (int axis = 0; axis < 3; axis++) {
                if (!use[axis]) continue;
                const Box* bb = all.bb[axis]; const uint32_t* bc = all.bc[axis];
                float rightArea[NB]; uint32_t rightCount[NB];
                Box acc; acc.reset(); uint32_t cnt = 0;
                for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k]); cnt += bc[k]; rightArea[k] = acc.area(); rightCount[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = 0; k < NB - 1; k++) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (cnt == 0 || rightCount[k + 1] == 0) continue;
                    const float cost = acc.area() * (float)cnt + rightArea[k + 1] * (float)rightCount[k + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = k; }
                }
            }
            const float leafCost = box.area() * (float)count;
            if (bestAxis >= 0 && (count > g_leafMax || bestCost + box.area() * g_travCost < leafCost)) {
                const float lo = cbox.lo[bestAxis], ext = cbox.hi[bestAxis] - lo;
                const float scale = 16.f / ext;
                auto it = std::partition(ids + first, ids + first + count, [&](uint32_t t) {
                    int k = (int)((cen[3 * (size_t)t + bestAxis] - lo) * scale);
                    k = std::min(15, std::max(0, k));
                    return k <= bestBin;
                });
                mid = (uint32_t)(it - ids);
                split = mid > first && mid < first + count;
            } else if (count <= g_leafMax) {
                return makeLeaf(first, count, box);
            }
        }
        if (!split) {
            if (count <= LM_MAX_LEAF && !forceMedian) return makeLeaf(first, count, box);
            if (count <= g_leafMax) return makeLeaf(first, count, box);
            int axis = 0;
            for (int k = 1; k < 3; k++) if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
            mid = first + count / 2;
            std::nth_element(ids + first, ids + mid, ids + first + count, [&](uint32_t a, uint32_t b) {
                const float ka = cen[3 * (size_t)a + axis], kb = cen[3 * (size_t)b + axis];
                return ka < kb || (ka == kb && a < b);
            });
        }
        const int self = (int)nodes.size();
        nodes.emplace_back();
        const Ref l = build(first, mid - first, depth + 1);
        const Ref r = build(mid, first + count - mid, depth + 1);
        setNode(self, l, r);
        Ref me; me.ref = self; me.box = box;
        return me;
    }

    void setNode(int idx, const Ref& l, const Ref& r)
    {
        LmNode& n = nodes[idx];
        auto P = [&](float v, float s) { return v + s * pad; };
        n.n0 = make_float4(P(l.box.lo[0], -1), P(l.box.hi[0], 1), P(l.box.lo[1], -1), P(l.box.hi[1], 1));
        n.n1 = make_float4(P(r.box.lo[0], -1), P(r.box.hi[0], 1), P(r.box.lo[1], -1), P(r.box.hi[1], 1));
        n.n2 = make_float4(P(l.box.lo[2], -1), P(l.box.hi[2], 1), P(r.box.lo[2], -1), P(r.box.hi[2], 1));
        n.ref = make_int4(l.ref, r.ref, 0, 0);
    }
};

}  // namespace

// CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a container can see 256 CPUs and be granted
// the time of 16; more threads than that only add time-slicing).
static unsigned usableCpus()
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    double quota = 0.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // cgroup v2: "<quota|max> <period>"
        char q[32]; double period = 0.0;
        if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) quota = atof(q) / period;
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        double q = 0.0, period = 100000.0;
        if (fscanf(g, "%lf", &q) != 1) q = 0.0;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &period) != 1) period = 100000.0; fclose(h); }
        if (q > 0.0 && period > 0.0) quota = q / period;
    }
    if (quota > 0.0) n = std::min(n, (unsigned)std::max(1.0, std::ceil(quota)));
    return std::max(1u, n);
}

// Octant slots of an 8-wide node (lm_layout.h LM_WIDTH): child i goes to the slot whose sign pattern best matches the offset of its centre
// from the centre of the node (slot bit k set = towards + along axis k), assigned greedily by the largest score sum_k +-(c_k - n_k); a ray
// then visits slot p ^ octant(direction), p = 0 .. 7, which is near to far for well-separated children (Ylitie, Karras, Laine 2017 use an
// auction for the same assignment; with at most 8 x 8 scores greedy is within noise of it).  Only the ORDER of visits depends on this.
// boxes: n x (lo.x hi.x lo.y hi.y lo.z hi.z).  4-wide: slot = child number.
static void lm_assign_slots(int n, const float (*b)[6], int* slotOf)
{
#if LM_WIDTH == 8
    float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { nlo[k] = std::min(nlo[k], b[i][2 * k]); nhi[k] = std::max(nhi[k], b[i][2 * k + 1]); }
    float d[8][3];
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
        const float ext = nhi[k] - nlo[k];
        d[i][k] = ext > 0.f ? (0.5f * (b[i][2 * k] + b[i][2 * k + 1]) - 0.5f * (nlo[k] + nhi[k])) / ext : 0.f;      // per-axis relative offset
    }
    bool usedC[8] = {false}, usedS[8] = {false};
    for (int it = 0; it < n; it++) {
        float best = -INFINITY; int bc = -1, bs = -1;
        for (int c = 0; c < n; c++) if (!usedC[c]) for (int s = 0; s < 8; s++) if (!usedS[s]) {
            const float score = ((s & 1) ? d[c][0] : -d[c][0]) + ((s & 2) ? d[c][1] : -d[c][1]) + ((s & 4) ? d[c][2] : -d[c][2]);
            if (score > best) { best = score; bc = c; bs = s; }
        }
        usedC[bc] = true; usedS[bs] = true; slotOf[bc] = bs;
    }
#else
    (void)b;
    for (int i = 0; i < n; i++) slotOf[i] = i;
#endif
}

void lm_build_bvh(const float* tris, uint32_t nTris, LmBvh* out)
{
    out->nodes.clear(); out->order.clear(); out->packets.clear();
    const bool timing = getenv("LUMEN_MI_BUILD_TIMING") != nullptr;
    auto tLast = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; const auto now = std::chrono::steady_clock::now(); fprintf(stderr, "[bvh] %-28s %.3f s\n", what, std::chrono::duration<double>(now - tLast).count()); tLast = now; };
    unsigned threads = std::max(1u, std::min(64u, usableCpus()));
    if (const char* e = getenv("LUMEN_MI_BUILD_THREADS")) threads = (unsigned)std::max(1, atoi(e));
    if (const char* e = getenv("LUMEN_MI_BVH_TRAV_COST")) g_travCost = (float)atof(e);
    if (const char* e = getenv("LUMEN_MI_BVH_SWEEP")) g_sweepBelow = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("LUMEN_MI_BVH_LEAF_MAX")) g_leafMax = (uint32_t)std::max(1, std::min((int)LM_MAX_LEAF, atoi(e)));
    std::vector<Box> tbox(nTris); std::vector<float> cen(3 * (size_t)nTris); std::vector<uint32_t> ids(nTris);
    std::vector<float> partMax(threads, 0.f);
    parallelChunks(nTris, threads, [&](unsigned tt, size_t lo, size_t hi) {
        float m = 0.f;
        for (size_t t = lo; t < hi; t++) {
            tbox[t].reset();
            for (int v = 0; v < 3; v++) {
                const float* p = tris + 9 * t + 3 * v;
                tbox[t].grow(p);
                for (int k = 0; k < 3; k++) m = std::max(m, std::fabs(p[k]));
            }
            for (int k = 0; k < 3; k++) cen[3 * t + k] = 0.5f * (tbox[t].lo[k] + tbox[t].hi[k]);
            ids[t] = (uint32_t)t;
        }
        partMax[tt] = m;
    });
    float maxAbs = 0.f;
    for (float m : partMax) maxAbs = std::max(maxAbs, m);
    lap("triangle boxes");
    Builder b;
    b.tbox = tbox.data(); b.cen = cen.data(); b.ids = ids.data(); b.threads = threads;
    b.pad = maxAbs * (1.0f / 32768.0f);
    out->pad = b.pad;
    std::vector<Task> tasks;
    if (threads > 1 && nTris >= 65536u) { b.tasks = &tasks; b.taskThreshold = std::max(4096u, nTris / (2u * threads)); }
    // absent child: marked by NaN boxes here; the quantised tree gives it the reference LM_REF_NONE
    Builder::Ref empty; empty.ref = ~0;
    for (int k = 0; k < 3; k++) { empty.box.lo[k] = NAN; empty.box.hi[k] = NAN; }
    if (nTris == 0) {
        b.nodes.emplace_back();
        b.setNode(0, empty, empty);
    } else {
        Builder::Ref root = b.build(0, nTris, 0);
        if (root.ref < 0) {                                    // whole scene fits one leaf: node 0 must still be an inner node
            b.nodes.emplace_back();
            b.setNode(0, root, empty);
        }
    }
    lap("top levels");
    // ---- subtrees in parallel, then concatenation in task order
    std::vector<Builder> sub(tasks.size());
    std::vector<Builder::Ref> subRoot(tasks.size());
    if (!tasks.empty()) {
        std::atomic<size_t> next{0};
        auto worker = [&] {
            for (size_t t; (t = next.fetch_add(1)) < tasks.size();) {
                Builder& w = sub[t];
                w.tbox = tbox.data(); w.cen = cen.data(); w.ids = ids.data(); w.pad = b.pad;
                subRoot[t] = w.build(tasks[t].first, tasks[t].count, tasks[t].depth);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::min<size_t>(threads, tasks.size()); t++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    lap("subtrees (parallel)");
    out->nodes = std::move(b.nodes);
    out->order = std::move(b.order);
    std::vector<int> rootRef(tasks.size());
    for (size_t t = 0; t < tasks.size(); t++) {
        const int nodeBase = (int)out->nodes.size();
        const uint32_t slotBase = (uint32_t)out->order.size();
        auto fix = [&](int ref) {
            if (ref >= 0) return ref + nodeBase;
            const uint32_t leaf = (uint32_t)(~ref);
            return ~(int)((((leaf >> 3) + slotBase) << 3) | (leaf & 7u));
        };
        for (LmNode n : sub[t].nodes) { n.ref.x = fix(n.ref.x); n.ref.y = fix(n.ref.y); out->nodes.push_back(n); }
        out->order.insert(out->order.end(), sub[t].order.begin(), sub[t].order.end());
        rootRef[t] = fix(subRoot[t].ref);
        b.maxDepth = std::max(b.maxDepth, sub[t].maxDepth);
        sub[t].nodes.clear(); sub[t].nodes.shrink_to_fit(); sub[t].order.clear(); sub[t].order.shrink_to_fit();
    }
    for (LmNode& n : out->nodes) {
        if (n.ref.x >= Builder::TASK_REF) n.ref.x = rootRef[n.ref.x - Builder::TASK_REF];
        if (n.ref.y >= Builder::TASK_REF) n.ref.y = rootRef[n.ref.y - Builder::TASK_REF];
    }
    lap("concatenation");
    out->maxDepth = b.maxDepth + 1;
    const size_t nSlots = out->order.size();
    out->packets.resize(nSlots + 1);
    parallelChunks(nSlots, threads, [&](unsigned, size_t lo, size_t hi) { for (size_t s = lo; s < hi; s++) out->packets[s] = lm_make_packet(tris + 9 * (size_t)out->order[s]); });
    memset(&out->packets[nSlots], 0, sizeof(LmTriPacket));               // sentinel packet: zero edge functions, never a hit

    lap("triangle packets");
    // ---- 16-bit quantisation relative to the (padded) scene box, rounded outward by LM_QUANT_MARGIN extra steps
    float smin[3] = {INFINITY, INFINITY, INFINITY}, smax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = 0; t < nTris; t++) for (int k = 0; k < 3; k++) { smin[k] = std::min(smin[k], tbox[t].lo[k]); smax[k] = std::max(smax[k], tbox[t].hi[k]); }
    for (int k = 0; k < 3; k++) {
        if (!(smin[k] <= smax[k])) { smin[k] = 0.f; smax[k] = 1.f; }
        const float m = 4.f * b.pad + 1e-6f * std::max(std::fabs(smin[k]), std::fabs(smax[k])) + 1e-30f;
        smin[k] -= m; smax[k] += m;
        out->qmin[k] = smin[k];
        out->qstep[k] = (smax[k] - smin[k]) / 65535.0f;
    }
    auto quant = [&](float lo, float hi, int k, uint32_t& packed) {
        const double inv = 1.0 / (double)out->qstep[k];
        long long ql = (long long)std::floor(((double)lo - (double)out->qmin[k]) * inv) - LM_QUANT_MARGIN;
        long long qh = (long long)std::ceil(((double)hi - (double)out->qmin[k]) * inv) + LM_QUANT_MARGIN;
        ql = std::max(0LL, std::min(65535LL, ql)); qh = std::max(0LL, std::min(65535LL, qh));
        packed = (uint32_t)ql | ((uint32_t)qh << 16);
    };
    // ---- collapse to LM_WIDTH-wide nodes: starting from a binary node's two children, repeatedly replace the inner child of
    // largest surface area by its own two children until there are LM_WIDTH (or only leaves are left)
    struct Child { int ref; float b[6]; uint32_t src; };      // padded box: lo.x hi.x lo.y hi.y lo.z hi.z; src = (binary node << 1) | side
    auto childrenOf = [&](int node, Child* c) {
        const LmNode& n = out->nodes[node];
        const float c0[6] = {n.n0.x, n.n0.y, n.n0.z, n.n0.w, n.n2.x, n.n2.y}, c1[6] = {n.n1.x, n.n1.y, n.n1.z, n.n1.w, n.n2.z, n.n2.w};
        int k = 0;
        if (c0[0] == c0[0]) { c[k].ref = n.ref.x; memcpy(c[k].b, c0, sizeof c0); c[k].src = (uint32_t)node << 1; k++; }
        if (c1[0] == c1[0]) { c[k].ref = n.ref.y; memcpy(c[k].b, c1, sizeof c1); c[k].src = ((uint32_t)node << 1) | 1u; k++; }
        return k;
    };
    auto areaOf = [](const Child& c) { const float dx = c.b[1] - c.b[0], dy = c.b[3] - c.b[2], dz = c.b[5] - c.b[4]; return dx * dy + dy * dz + dz * dx; };
    // The collapse is a top-down walk whose subtrees are independent.  It runs sequentially until enough subtrees are
    // pending, then every pending subtree is (1) counted and (2) written by a worker into its own index range — the
    // prefix sum over the counts keeps the node numbering independent of thread scheduling.
    struct Work { int node2; int node4; uint32_t stack; uint32_t depth; };
    std::vector<uint32_t> depthOf;
    out->maxStack = 1;
    // expands one work item: children + the number of 4-wide children it creates
    auto expand = [&](const Work& w, Child* c) {
        int n = childrenOf(w.node2, c);
        while (n < LM_WIDTH) {
            int best = -1; float bestArea = -1.f;
            for (int i = 0; i < n; i++) if (c[i].ref >= 0) { const float a = areaOf(c[i]); if (a > bestArea) { bestArea = a; best = i; } }
            if (best < 0) break;
            Child g[2];
            const int m = childrenOf(c[best].ref, g);
            if (n - 1 + m > LM_WIDTH) break;
            c[best] = g[0];
            if (m > 1) c[n++] = g[1];
        }
        return n;
    };
    // writes node w.node4 and pushes its inner children (indices taken from `nextId`) on `stack`
    auto emit = [&](const Work& w, std::vector<Work>& stack, int& nextId, uint32_t& maxStack) {
        Child c[LM_WIDTH];
        const int n = expand(w, c);
        const uint32_t stackBelow = w.stack + (uint32_t)(n > 0 ? n - 1 : 0);
        maxStack = std::max(maxStack, stackBelow + 1u);
        LmNodeW q;
        int slotOf[LM_WIDTH];
        float boxes[LM_WIDTH][6];
        for (int i = 0; i < n; i++) memcpy(boxes[i], c[i].b, sizeof boxes[i]);
        lm_assign_slots(n, boxes, slotOf);
        for (int s = 0; s < LM_WIDTH; s++) q.c[s] = make_uint4(LM_BOX_NONE, LM_BOX_NONE, LM_BOX_NONE, (uint32_t)LM_REF_NONE);
        for (int i = 0; i < n; i++) {
            int ref = c[i].ref;
            if (ref >= 0) {
                const int id4 = nextId++;
                stack.push_back({ref, id4, stackBelow, w.depth + 1});
                depthOf[id4] = w.depth + 1;
                ref = id4;
            }
            q.c[slotOf[i]] = make_uint4(c[i].src, 0u, 0u, (uint32_t)ref);        // .x = where the box comes from; quantised below
        }
        out->nodesW[w.node4] = q;
    };
    out->nodesW.assign(1, LmNodeW{});
    depthOf.assign(1, 0);
    std::vector<Work> work;
    work.push_back({0, 0, 0, 0});
    int nextId = 1;
    while (!work.empty() && (threads <= 1 || work.size() < 8u * threads)) {   // sequential part (the whole tree when single-threaded)
        const Work w = work.back(); work.pop_back();
        if (out->nodesW.size() < (size_t)nextId + LM_WIDTH) { out->nodesW.resize(2 * (size_t)nextId + LM_WIDTH); depthOf.resize(2 * (size_t)nextId + LM_WIDTH); }
        emit(w, work, nextId, out->maxStack);
    }
    if (!work.empty()) {
        std::vector<uint32_t> cnt(work.size(), 0);
        auto countSubtree = [&](const Work& root) {                 // 4-wide nodes strictly below `root`
            uint32_t total = 0;
            std::vector<int> st{root.node2};
            while (!st.empty()) {
                Work w{st.back(), 0, 0, 0}; st.pop_back();
                Child c[LM_WIDTH];
                const int n = expand(w, c);
                for (int i = 0; i < n; i++) if (c[i].ref >= 0) { total++; st.push_back(c[i].ref); }
            }
            return total;
        };
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; t++) pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < work.size();) cnt[i] = countSubtree(work[i]); });
        for (auto& th : pool) th.join();
        std::vector<int> base(work.size());
        for (size_t i = 0; i < work.size(); i++) { base[i] = nextId; nextId += (int)cnt[i]; }
        out->nodesW.resize((size_t)nextId); depthOf.resize((size_t)nextId);
        std::vector<uint32_t> tmax(threads, 1u);
        next = 0; pool.clear();
        for (unsigned t = 0; t < threads; t++) pool.emplace_back([&, t] {
            std::vector<Work> st;
            uint32_t localMax = 1u;                                 // (a shared array element here would be falsely shared)
            for (size_t i; (i = next.fetch_add(1)) < work.size();) {
                int id = base[i];
                st.clear(); st.push_back(work[i]);
                while (!st.empty()) { const Work w = st.back(); st.pop_back(); emit(w, st, id, localMax); }
            }
            tmax[t] = localMax;
        });
        for (auto& th : pool) th.join();
        for (uint32_t m : tmax) out->maxStack = std::max(out->maxStack, m);
    }
    out->nodesW.resize((size_t)nextId);
    depthOf.resize((size_t)nextId);
    lap("collapse");
    parallelChunks(out->nodesW.size(), threads, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) for (int k = 0; k < LM_WIDTH; k++) {
            uint4& ch = out->nodesW[i].c[k];
            if ((int)ch.w == LM_REF_NONE) continue;
            const LmNode& n = out->nodes[ch.x >> 1];
            const float c0[6] = {n.n0.x, n.n0.y, n.n0.z, n.n0.w, n.n2.x, n.n2.y}, c1[6] = {n.n1.x, n.n1.y, n.n1.z, n.n1.w, n.n2.z, n.n2.w};
            const float* bx = (ch.x & 1u) ? c1 : c0;
            uint32_t p[3];
            for (int a = 0; a < 3; a++) quant(bx[2 * a], bx[2 * a + 1], a, p[a]);
            ch.x = p[0]; ch.y = p[1]; ch.z = p[2];
        }
    });
    lap("quantise");
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


// ---------------------------------------------------------------------------------------------------------------------
// instance-level assembly (bvh.h)
// ---------------------------------------------------------------------------------------------------------------------
void lm_assemble_bvh(const LmInstanceRef* inst, uint32_t nInst, LmBvh* out)
{
    *out = LmBvh();
    struct Top { int child[LM_WIDTH]; };                     // per slot: >= 0 top node, < 0 ~instance, LM_REF_NONE absent
    std::vector<Top> top;
    std::vector<uint32_t> ids(nInst);
    for (uint32_t i = 0; i < nInst; i++) ids[i] = i;
    auto centre = [&](uint32_t i, int k) { return 0.5f * (inst[i].box[k] + inst[i].box[3 + k]); };
    // splits ids[lo, hi) at the median of the centroids along their longest axis (ties by instance number: deterministic)
    auto splitHalf = [&](uint32_t lo, uint32_t hi) {
        float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t k = lo; k < hi; k++) for (int a = 0; a < 3; a++) { cl[a] = std::min(cl[a], centre(ids[k], a)); ch[a] = std::max(ch[a], centre(ids[k], a)); }
        int axis = 0;
        for (int a = 1; a < 3; a++) if (ch[a] - cl[a] > ch[axis] - cl[axis]) axis = a;
        const uint32_t mid = lo + (hi - lo) / 2u;
        std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t a, uint32_t b) {
            const float ca = centre(a, axis), cb = centre(b, axis);
            return ca < cb || (ca == cb && a < b);
        });
        return mid;
    };
    struct Range { uint32_t lo, hi; };
    // a top node = its range halved at medians until there are LM_WIDTH groups (or only single instances); recursion depth log_W(nInst)
    std::function<int(uint32_t, uint32_t)> build = [&](uint32_t lo, uint32_t hi) -> int {
        if (hi - lo == 1u) return ~(int)ids[lo];
        const int me = (int)top.size();
        top.push_back(Top{});
        std::vector<Range> g{Range{lo, hi}};
        while ((int)g.size() < LM_WIDTH) {
            size_t pick = g.size(); uint32_t best = 1u;
            for (size_t k = 0; k < g.size(); k++) if (g[k].hi - g[k].lo > best) { best = g[k].hi - g[k].lo; pick = k; }
            if (pick == g.size()) break;
            const uint32_t m = splitHalf(g[pick].lo, g[pick].hi);
            const Range a{g[pick].lo, m}, b{m, g[pick].hi};
            g[pick] = a; g.insert(g.begin() + (long)pick + 1, b);
        }
        float boxes[LM_WIDTH][6]; int slotOf[LM_WIDTH];
        for (size_t k = 0; k < g.size(); k++) {
            float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = g[k].lo; i < g[k].hi; i++) for (int a = 0; a < 3; a++) { lo3[a] = std::min(lo3[a], inst[ids[i]].box[a]); hi3[a] = std::max(hi3[a], inst[ids[i]].box[3 + a]); }
            for (int a = 0; a < 3; a++) { boxes[k][2 * a] = lo3[a]; boxes[k][2 * a + 1] = hi3[a]; }
        }
        lm_assign_slots((int)g.size(), boxes, slotOf);
        Top t; for (int s = 0; s < LM_WIDTH; s++) t.child[s] = LM_REF_NONE;
        for (size_t k = 0; k < g.size(); k++) { const int c = build(g[k].lo, g[k].hi); t.child[slotOf[k]] = c; }
        top[(size_t)me] = t;
        return me;
    };
    if (nInst == 0) { out->nodesW.assign(1, LmNodeW{}); for (auto& c : out->nodesW[0].c) c = make_uint4(LM_BOX_NONE, LM_BOX_NONE, LM_BOX_NONE, (uint32_t)LM_REF_NONE); }
    else if (nInst > 1) build(0, nInst);
    const uint32_t T = (uint32_t)top.size();
    std::vector<uint32_t> nodeBase(nInst), slotBase(nInst);
    uint32_t nNodes = T, nSlots = 0;
    for (uint32_t i = 0; i < nInst; i++) { nodeBase[i] = nNodes; slotBase[i] = nSlots; nNodes += (uint32_t)inst[i].mesh->nodesW.size(); nSlots += (uint32_t)inst[i].mesh->order.size(); }
    if (nInst) out->nodesW.resize(nNodes);
    out->order.resize(nSlots);
    for (uint32_t t = 0; t < T; t++) {
        LmNodeW q;
        for (int k = 0; k < LM_WIDTH; k++) {
            const int c = top[t].child[k];
            if (c == LM_REF_NONE) { q.c[k] = make_uint4(LM_BOX_NONE, LM_BOX_NONE, LM_BOX_NONE, (uint32_t)LM_REF_NONE); continue; }
            q.c[k] = make_uint4(0u, 0u, 0u, (uint32_t)(c >= 0 ? c : (int)nodeBase[(uint32_t)(~c)]));
        }
        out->nodesW[t] = q;
    }
    for (uint32_t i = 0; i < nInst; i++) {
        const LmBvh& m = *inst[i].mesh;
        for (size_t n = 0; n < m.nodesW.size(); n++) {
            LmNodeW q = m.nodesW[n];
            for (auto& c : q.c) {
                const int ref = (int)c.w;
                if (ref == LM_REF_NONE) continue;
                if (ref >= 0) c.w = (uint32_t)(ref + (int)nodeBase[i]);
                else { const uint32_t leaf = (uint32_t)(~ref); c.w = (uint32_t)(~(int)((((leaf >> 3) + slotBase[i]) << 3) | (leaf & 7u))); }
            }
            out->nodesW[nodeBase[i] + n] = q;
        }
        for (size_t s = 0; s < m.order.size(); s++) out->order[slotBase[i] + s] = inst[i].triBase + m.order[s];
    }
    out->packets.assign((size_t)nSlots + 1, LmTriPacket{});                   // refit_tris writes the packets; the sentinel stays zero
    // depth of every node, worst-case stack occupancy (the rule of lm_build_bvh's collapse), refit levels deepest first
    std::vector<uint32_t> depthOf(out->nodesW.size(), 0);
    struct Item { uint32_t node, depth, stack; };
    std::vector<Item> st{{0u, 0u, 0u}};
    out->maxStack = 1; uint32_t maxDepth = 0;
    while (!st.empty()) {
        const Item w = st.back(); st.pop_back();
        depthOf[w.node] = w.depth; maxDepth = std::max(maxDepth, w.depth);
        uint32_t present = 0;
        for (const auto& c : out->nodesW[w.node].c) present += (int)c.w != LM_REF_NONE;
        const uint32_t below = w.stack + (present ? present - 1u : 0u);
        out->maxStack = std::max(out->maxStack, below + 1u);
        for (const auto& c : out->nodesW[w.node].c) if ((int)c.w >= 0 && (int)c.w != LM_REF_NONE) st.push_back({c.w, w.depth + 1u, below});
    }
    out->maxDepth = maxDepth + 1u;
    std::vector<uint32_t> count(maxDepth + 2u, 0);
    for (uint32_t d : depthOf) count[maxDepth - d + 1u]++;             // level 0 = deepest
    out->levelStart.assign(maxDepth + 2u, 0);
    for (uint32_t l = 1; l < maxDepth + 2u; l++) out->levelStart[l] = out->levelStart[l - 1] + count[l];
    out->levelNodes.resize(out->nodesW.size());
    std::vector<uint32_t> fill(out->levelStart.begin(), out->levelStart.end() - 1);
    for (uint32_t n = 0; n < (uint32_t)out->nodesW.size(); n++) out->levelNodes[fill[maxDepth - depthOf[n]]++] = n;
}
