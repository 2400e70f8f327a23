// bvh_gpu.hip — acceleration-structure build ON THE DEVICE for new geometry (tuning key "gpu_build").
//
// Reference: OptixWrapper::BuildGeometryAccelerationStructure / BuildInstanceAccelerationStructure (LumenPT/src/Framework/OptixWrapper.cpp:46-78,80-131) build on the GPU
// (optixAccelBuild, closed).  Here: a linear BVH in five steps, all on the device, from the scene tables the renderer already keeps there (instance table, vertex and
// index pools) — no world-space triangle soup is made on the host:
//   1. world-space box and centroid of every triangle (the transform arithmetic of lm_k_refit_tris), scene box of the centroids
//   2. 63-bit Morton key of the centroid (21 bits per axis), radix sort of (key, triangle) pairs (hipCUB)
//   3. Karras 2012 binary radix tree over the sorted keys (ties broken by position, so equal keys still split), parents, leaf ranges
//   4. bottom-up boxes of the binary nodes (second arrival at a node merges its children)
//   5. top-down collapse to the LM_WIDTH-wide tree the kernels read, one launch per depth level: a node's two children are expanded by replacing the inner
//      child of largest surface area until there are LM_WIDTH (what bvh.cpp's collapse does with the SAH tree); a subtree of at most LM_GPU_LEAF triangles
//      becomes one leaf — its triangles are contiguous in the sorted order.  Depth per node and the worst-case stack need come out of the same walk.
// Output = TOPOLOGY: child references, triangle order, depth levels.  Child boxes and Woop packets are computed by the refit kernels the renderer already
// runs after every topology change (kernels.hip lm_k_refit_tris / _quant / _level: bit-identical packets to the host builder, outward-rounded 16-bit boxes).
// Hit records do not depend on the tree (closest t, ties by triangle number), so every parity test holds unchanged under this builder; what changes is how
// many nodes a ray visits — an LBVH is a worse tree than the host's binned-SAH one (A/B: profiles/r04_gpu_build_ab.txt), which is why SAH stays the default
// for a scene's first build and this one is for geometry that has to be traceable NOW.
#include "bvh.h"
#include "renderer_state.h"
#include <hipcub/hipcub.hpp>

#ifndef LM_GPU_LEAF
#define LM_GPU_LEAF 4u          // triangles per leaf (the format allows LM_MAX_LEAF = 8; the SAH builder's default maximum is 4 as well)
#endif

namespace {

struct Bin2 { int left, right; };                            // child >= 0: inner node; < 0: ~sorted slot
__device__ __forceinline__ uint32_t ordf(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float unordf(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// 1. per input triangle: world-space box + centroid; scene box of the centroids (ordered-uint atomics)
__global__ void k_tri_boxes(const LmEntry* __restrict__ entries, const float4* __restrict__ verts, const uint32_t* __restrict__ indices, const uint2* __restrict__ triIn,
                            uint32_t n, float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* cbounds)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (t < n) {
        const uint2 id = triIn[t];
        const LmEntry e = entries[id.x];
        for (int k = 0; k < 3; k++) {
            const uint32_t vi = indices[e.idxBase + 3u * id.y + (uint32_t)k];
            const float4 p = verts[3u * (e.vertBase + vi)];
            const float w[3] = {e.m[0] * p.x + e.m[1] * p.y + e.m[2] * p.z + e.m[3] * 1.f, e.m[4] * p.x + e.m[5] * p.y + e.m[6] * p.z + e.m[7] * 1.f,
                                e.m[8] * p.x + e.m[9] * p.y + e.m[10] * p.z + e.m[11] * 1.f};
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], w[a]); hi[a] = fmaxf(hi[a], w[a]); }
        }
        boxLo[t] = make_float4(lo[0], lo[1], lo[2], 0.f); boxHi[t] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    // scene box of the centroids: reduced over the wavefront and the block first, ONE set of atomics per block (round 6: six atomics per TRIANGLE on six words of one
    // cache line were most of this kernel — 60 M of them for the 10 M-triangle scene; min / max are order independent, so the bounds are the same bits)
    __shared__ float s_red[6 * 16];
    float cmin[3], cmax[3];
    for (int a = 0; a < 3; a++) {
        const float c = 0.5f * (lo[a] + hi[a]);
        const bool ok = t < n && c == c;
        cmin[a] = ok ? c : INFINITY; cmax[a] = ok ? c : -INFINITY;
        for (int o = 32; o > 0; o >>= 1) { cmin[a] = fminf(cmin[a], __shfl_xor(cmin[a], o)); cmax[a] = fmaxf(cmax[a], __shfl_xor(cmax[a], o)); }
    }
    const uint32_t wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63u) >> 6;
    if ((threadIdx.x & 63u) == 0u) for (int a = 0; a < 3; a++) { s_red[6u * wave + a] = cmin[a]; s_red[6u * wave + 3 + a] = cmax[a]; }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t k = threadIdx.x;
        float v = s_red[k];
        for (uint32_t w = 1; w < nWaves; w++) v = k < 3u ? fminf(v, s_red[6u * w + k]) : fmaxf(v, s_red[6u * w + k]);
        if (k < 3u) { if (v != INFINITY) atomicMin(cbounds + k, ordf(v)); }
        else if (v != -INFINITY) atomicMax(cbounds + k, ordf(v));
    }
}
__device__ __forceinline__ unsigned long long spread21(unsigned long long v)      // 21 bits -> every third bit
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
// 2. Morton keys
__global__ void k_morton(const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, uint32_t n, const uint32_t* __restrict__ cbounds, unsigned long long* keys, uint32_t* vals)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float4 a = boxLo[t], b = boxHi[t];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    unsigned long long key = 0ull;
    for (int k = 0; k < 3; k++) {
        const float lo = unordf(cbounds[k]), hi = unordf(cbounds[3 + k]);
        const float ext = hi - lo;
        float rel = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        rel = rel == rel ? fminf(fmaxf(rel, 0.f), 1.f) : 0.f;
        const unsigned long long q = (unsigned long long)fminf(rel * 2097152.0f, 2097151.0f);
        key |= spread21(q) << k;
    }
    keys[t] = key; vals[t] = t;
}
// 3. binary radix tree (Karras 2012).  delta(i, j): common prefix of keys i and j; equal keys fall back on the positions themselves
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}
__global__ void k_radix_tree(const unsigned long long* __restrict__ keys, int n, Bin2* nodes, int* parentInner, int* parentLeaf, uint2* range)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = first == gamma ? ~gamma : gamma, right = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    nodes[i] = Bin2{left, right};
    range[i] = make_uint2((uint32_t)first, (uint32_t)last);
    if (left >= 0) parentInner[left] = i; else parentLeaf[~left] = i;
    if (right >= 0) parentInner[right] = i; else parentLeaf[~right] = i;
    if (i == 0) parentInner[0] = -1;
}
// 4. bottom-up boxes: one thread per leaf walks up; the second thread to arrive at a node merges the children's boxes
__global__ void k_fit(const Bin2* __restrict__ nodes, const int* __restrict__ parentInner, const int* __restrict__ parentLeaf, const uint32_t* __restrict__ sortedTri,
                      const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, int n, float4* nodeLo, float4* nodeHi, uint32_t* arrived)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int p = parentLeaf[s];
    while (p >= 0) {
        __threadfence();                                  // release: this thread's box is visible before it announces itself
        if (atomicAdd(arrived + p, 1u) == 0u) return;
        __threadfence();                                  // acquire: the sibling's box (written by another wave, maybe on another CU) is read after the counter, not from a
                                                          // line this CU's vector L1 already held — a stale area would only change which child the collapse expands, but that
                                                          // made the tree's shape differ from run to run
        const Bin2 b = nodes[p];
        const float4 alo = b.left >= 0 ? nodeLo[b.left] : boxLo[sortedTri[~b.left]], ahi = b.left >= 0 ? nodeHi[b.left] : boxHi[sortedTri[~b.left]];
        const float4 blo = b.right >= 0 ? nodeLo[b.right] : boxLo[sortedTri[~b.right]], bhi = b.right >= 0 ? nodeHi[b.right] : boxHi[sortedTri[~b.right]];
        nodeLo[p] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.f);
        nodeHi[p] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
        p = parentInner[p];
    }
}
// 5. one depth level of the collapse.  Work item: binary inner node -> wide node id, with the stack occupancy above it.
struct Work { int node2, node4; uint32_t stackAbove; };
__device__ __forceinline__ float areaOf(const float4& lo, const float4& hi) { const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z; return dx * dy + dy * dz + dz * dx; }
__global__ void k_collapse_level(const Work* __restrict__ in, uint32_t nIn, Work* out, uint32_t* counters /* [0] wide nodes, [1] next queue length, [2] max stack */,
                                 const Bin2* __restrict__ nodes, const uint2* __restrict__ range, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
                                 LmNodeW* wide, uint32_t* depthOf, uint32_t depth)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nIn) return;
    const Work w = in[i];
    int c[LM_WIDTH];                       // child: >= 0 binary inner node that stays inner; < 0 leaf reference already in the kernels' format
    float area[LM_WIDTH];
    int n = 0;
    auto add = [&](int child) {            // a binary child: a single triangle, a small subtree (one leaf), or an inner node
        if (child < 0) { c[n] = ~(int)(((uint32_t)(~child) << 3) | 0u); area[n] = -1.f; n++; return; }
        const uint2 r = range[child];
        const uint32_t cnt = r.y - r.x + 1u;
        if (cnt <= LM_GPU_LEAF) { c[n] = ~(int)((r.x << 3) | (cnt - 1u)); area[n] = -1.f; n++; return; }
        c[n] = child; area[n] = areaOf(nodeLo[child], nodeHi[child]); n++;
    };
    { const Bin2 b = nodes[w.node2]; add(b.left); add(b.right); }
    while (n < LM_WIDTH) {
        int best = -1; float bestArea = -1.f;
        for (int k = 0; k < n; k++) if (c[k] >= 0 && area[k] > bestArea) { bestArea = area[k]; best = k; }
        if (best < 0) break;
        const Bin2 b = nodes[c[best]];
        // the expanded child is replaced by its left child in place and the right one is appended
        const int keepN = n;
        n = best; add(b.left);
        const int c0 = c[best]; const float a0 = area[best];
        n = keepN; add(b.right);
        c[best] = c0; area[best] = a0;
    }
    const uint32_t stackBelow = w.stackAbove + (uint32_t)(n - 1);
    atomicMax(counters + 2, stackBelow + 1u);
    LmNodeW q;
    for (int k = 0; k < LM_WIDTH; k++) q.c[k] = make_uint4(LM_BOX_NONE, LM_BOX_NONE, LM_BOX_NONE, (uint32_t)LM_REF_NONE);
    for (int k = 0; k < n; k++) {
        int ref = c[k];
        if (ref >= 0) {
            const uint32_t id4 = atomicAdd(counters + 0, 1u);
            const uint32_t slot = atomicAdd(counters + 1, 1u);
            out[slot] = Work{ref, (int)id4, stackBelow};
            depthOf[id4] = depth + 1u;
            ref = (int)id4;
        }
        q.c[k] = make_uint4(0u, 0u, 0u, (uint32_t)ref);            // boxes: the refit kernels
    }
    wide[w.node4] = q;
}

struct Dev {
    std::vector<void*> all;
    template <class T> T* get(size_t count) { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr; all.push_back(p); return (T*)p; }
    ~Dev() { for (void* p : all) (void)hipFree(p); }
};

}  // namespace

// Returns 0, or non-zero when this builder does not apply (the caller then uses the host SAH builder): an allocation failed, fewer than two triangles, an
// LM_WIDTH other than 4, or a tree whose worst-case traversal stack would exceed LM_STACK_DEPTH (degenerate inputs: the radix tree is not depth-bounded).
int lm_build_bvh_gpu(hipStream_t st, const LmEntry* dEntries, const float4* dVerts, const uint32_t* dIndices, const uint2* dTriIn, uint32_t nTris, LmBvh* out)
{
#if LM_WIDTH != 4
    (void)st; (void)dEntries; (void)dVerts; (void)dIndices; (void)dTriIn; (void)nTris; (void)out;
    return 1;
#else
    if (nTris < 2u || nTris >= (1u << 28)) return 1;
    const int n = (int)nTris;
    Dev d;
    float4* boxLo = d.get<float4>(nTris); float4* boxHi = d.get<float4>(nTris);
    uint32_t* cbounds = d.get<uint32_t>(8);
    unsigned long long* keys = d.get<unsigned long long>(nTris); unsigned long long* keysOut = d.get<unsigned long long>(nTris);
    uint32_t* vals = d.get<uint32_t>(nTris); uint32_t* valsOut = d.get<uint32_t>(nTris);
    Bin2* nodes = d.get<Bin2>(nTris); int* parentInner = d.get<int>(nTris); int* parentLeaf = d.get<int>(nTris); uint2* range = d.get<uint2>(nTris);
    float4* nodeLo = d.get<float4>(nTris); float4* nodeHi = d.get<float4>(nTris); uint32_t* arrived = d.get<uint32_t>(nTris);
    // wide nodes: every wide node has at least two children and every leaf at least one triangle: fewer than nTris of them
    LmNodeW* wide = d.get<LmNodeW>(nTris); uint32_t* depthOf = d.get<uint32_t>(nTris);
    Work* q0 = d.get<Work>(nTris); Work* q1 = d.get<Work>(nTris);
    uint32_t* counters = d.get<uint32_t>(4);
    if (!boxLo || !boxHi || !cbounds || !keys || !keysOut || !vals || !valsOut || !nodes || !parentInner || !parentLeaf || !range || !nodeLo || !nodeHi || !arrived || !wide || !depthOf || !q0 || !q1 || !counters) return 1;
    const uint32_t cb[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    if (hipMemcpyAsync(cbounds, cb, sizeof cb, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    const unsigned B = 256, G = (nTris + B - 1) / B;
    hipLaunchKernelGGL(k_tri_boxes, dim3(G), dim3(B), 0, st, dEntries, dVerts, dIndices, dTriIn, nTris, boxLo, boxHi, cbounds);
    hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, st, boxLo, boxHi, nTris, cbounds, keys, vals);
    size_t tempBytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keys, keysOut, vals, valsOut, n, 0, 63, st) != hipSuccess) return 1;
    void* temp = d.get<uint8_t>(tempBytes);
    if (!temp || hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keys, keysOut, vals, valsOut, n, 0, 63, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_radix_tree, dim3(G), dim3(B), 0, st, keysOut, n, nodes, parentInner, parentLeaf, range);
    if (hipMemsetAsync(arrived, 0, (size_t)nTris * 4, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_fit, dim3(G), dim3(B), 0, st, nodes, parentInner, parentLeaf, valsOut, boxLo, boxHi, n, nodeLo, nodeHi, arrived);
    // collapse, level by level (the queue length comes back to the host once per level: a 4-byte read)
    const uint32_t c0[4] = {1u, 0u, 1u, 0u};                    // wide node 0 = the root, already allocated
    const Work root{0, 0, 0u};
    if (hipMemcpyAsync(counters, c0, sizeof c0, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(q0, &root, sizeof root, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(depthOf, 0, 4, st) != hipSuccess) return 1;
    uint32_t nIn = 1u, depth = 0u;
    Work* in = q0; Work* next = q1;
    while (nIn) {
        hipLaunchKernelGGL(k_collapse_level, dim3((nIn + B - 1) / B), dim3(B), 0, st, in, nIn, next, counters, nodes, range, nodeLo, nodeHi, wide, depthOf, depth);
        uint32_t len = 0;
        if (hipMemcpyAsync(&len, counters + 1, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemsetAsync(counters + 1, 0, 4, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
        nIn = len; std::swap(in, next); ++depth;
        if (depth > 4096u) return 1;
    }
    uint32_t fin[4];
    if (hipMemcpy(fin, counters, sizeof fin, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const uint32_t nWide = fin[0];
    if (fin[2] > (uint32_t)LM_STACK_DEPTH) return 2;              // a tree the traversal stack cannot hold: the host builder bounds its depth, this one does not
    *out = LmBvh();
    out->nodesW.resize(nWide); out->order.resize(nTris);
    std::vector<uint32_t> depthHost(nWide);
    if (hipMemcpy(out->nodesW.data(), wide, (size_t)nWide * sizeof(LmNodeW), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(out->order.data(), valsOut, (size_t)nTris * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(depthHost.data(), depthOf, (size_t)nWide * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    out->maxStack = fin[2];
    out->maxDepth = depth;                                        // levels of the wide tree (the binary depth is not tracked; the stack bound above is what matters)
    uint32_t maxD = 0;
    for (uint32_t dd : depthHost) maxD = std::max(maxD, dd);
    out->levelStart.assign(maxD + 2, 0);
    for (uint32_t dd : depthHost) out->levelStart[maxD - dd + 1]++;
    for (uint32_t l = 0; l <= maxD; l++) out->levelStart[l + 1] += out->levelStart[l];
    out->levelNodes.resize(nWide);
    std::vector<uint32_t> fill(out->levelStart.begin(), out->levelStart.end() - 1);
    for (uint32_t i = 0; i < nWide; i++) out->levelNodes[fill[maxD - depthHost[i]]++] = i;
    return 0;
#endif
}
