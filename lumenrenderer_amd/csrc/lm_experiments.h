// lm_experiments.h — variants that were BUILT, MEASURED AND NOT ADOPTED, kept compilable (each has a recorded A/B under profiles/ and tests that run it) but out of the
// way of the hot path's two files.  Included by lm_traverse.h / kernels.hip at the point of use with LM_EXPERIMENTS_PART set:
//   1  the 8-wide node step (-DLM_WIDTH=8; profiles/r03_wide_ab.txt: -5 ... -15 %)
//   2  node fetch / evaluation in two halves (-DLM_NODE_PIPELINE=1; profiles/r02_node_pipeline_ab.txt: -5 %)
//   3  the octant-ordered block append (-DLM_APPEND_OCTANT=1; profiles/r05_append_octant_ab.txt: -0.4 %)
//   4  the repacking path tail (tuning key tail_repack; profiles/r04_tail_repack_ab.txt: -1.4 %) — compiled in every build, because the key is part of the tested surface
// What stays inline in lm_traverse.h / kernels.hip are the switches that are a few lines inside a loop (LM_LEAF_PAIR, LM_SPECULATE, LM_SLAB_PERM 0 / 1, the 8-wide packet
// branch) and the kernels behind tested tuning keys (spatial_lds, the persistent pick).
#if LM_EXPERIMENTS_PART == 1
// 8-wide node step.  The eight child records are fetched in the ray's VISITING order — record p from octant slot p ^ octant(direction): a
// per-lane address inside the node's one 128-byte line — so everything behind the loads is compile-time ordered: p = 0 is the child
// nearest along the ray, p = 7 the farthest.  No sorting network: every hit child is pushed far to near, and the nearest, which ends on top,
// is taken back at once.  The pushes of the common case are unconditional LDS stores whose stack pointer only advances for a hit (no
// branches); a lane whose LDS share of the stack could overflow within this step takes the branching lm_push path.
template <bool ANY>
__device__ __forceinline__ int lm_node_step(const LmScene& sc, int cur, const LmRayQ& rq, float tmin, float hitT, const LmStack& stack, int& sp,
                                            const lm_lds_u4* top, uint32_t* boxes = nullptr)
{
    uint4 q[8];
#if LM_TOP_NODES
    if (cur >= LM_TOP_BASE) {                                    // (only kernels that staged the table ever hold such a reference)
        const uint32_t off = ((uint32_t)(cur - LM_TOP_BASE) << 7) | rq.oct;
        typedef __attribute__((address_space(3))) char lm_lds_char;
        const lm_lds_char* tb = (const lm_lds_char*)top;
#pragma unroll
        for (int p = 0; p < 8; p++) q[p] = lm_lds_read4((const lm_lds_u4*)(tb + (off ^ ((uint32_t)p << 4))));
    } else
#endif
    {
        const uint32_t off = ((uint32_t)cur << 7) | rq.oct;
        const char* nb = (const char*)sc.nodes;
#pragma unroll
        for (int p = 0; p < 8; p++) q[p] = *(const uint4*)(nb + (off ^ ((uint32_t)p << 4)));
    }
    if (boxes) { for (int p = 0; p < 8; p++) *boxes += (int)q[p].w != LM_REF_NONE; }      // counting build
    uint32_t k[8];
#pragma unroll
    for (int p = 0; p < 8; p++) lm_slab(q[p], rq, tmin, hitT, k[p]);
    int next = LM_REF_NONE;
    if (sp + 8 <= LM_STACK_LDS) {
        lm_lds_int* sl = stack.lds + sp * LM_BLOCK;
#pragma unroll
        for (int p = 7; p >= 0; p--) {
            const bool h = k[p] != 0xffffffffu;
            *sl = (int)q[p].w;                                   // overwritten by the next store unless this child is hit
            sl += h ? LM_BLOCK : 0; sp += h ? 1 : 0;
            next = h ? (int)q[p].w : next;
        }
    } else {
#pragma unroll
        for (int p = 7; p >= 0; p--) if (k[p] != 0xffffffffu) { lm_push(stack, sp, (int)q[p].w); next = (int)q[p].w; }
    }
    if (next == LM_REF_NONE) return sp == 0 ? LM_REF_NONE : lm_pop(stack, sp);
    --sp;                                                        // the nearest hit child is on top: continue with it
    return next;
}
#elif LM_EXPERIMENTS_PART == 2
// The same step in two halves for the queue kernels (LM_NODE_PIPELINE): the record fetch, and the evaluation, which issues the fetch of the
// child it continues with as soon as that child is known — after three of the five comparators (closest hit) or the first-hit select
// (any hit) — so that the rest of the ordering and the stack pushes run under the load instead of in front of it.  `pre` tells the
// caller that q0..q3 already hold the records of the returned node.  Visiting order, stack contents and results are those of lm_node_step.
// Measured and NOT used: the records in flight stay live across the pushes, which costs 12 more VGPRs — closest hit spills at eight
// waves per SIMD (200 -> 400 us) or runs with six (230 us), visibility 228 -> 300 us, NEE shadow 91 -> 106 us, frame -5 %
// (profiles/r02_node_pipeline_ab.txt): occupancy hides the load better than the overlap does.
__device__ __forceinline__ void lm_node_fetch(const LmScene& sc, int cur, const lm_lds_u4* top, uint4& q0, uint4& q1, uint4& q2, uint4& q3)
{
#if LM_TOP_NODES
    if (cur >= LM_TOP_BASE) {                                    // (only kernels that staged the table ever hold such a reference)
        const lm_lds_u4* nd = top + 4u * (uint32_t)(cur - LM_TOP_BASE);
        q0 = lm_lds_read4(nd); q1 = lm_lds_read4(nd + 1); q2 = lm_lds_read4(nd + 2); q3 = lm_lds_read4(nd + 3);
    } else
#endif
    {
        const uint4* nd = sc.nodes[cur].c;
        q0 = nd[0]; q1 = nd[1]; q2 = nd[2]; q3 = nd[3];
    }
}
template <bool ANY>
__device__ __forceinline__ int lm_node_eval(const LmScene& sc, uint4& q0, uint4& q1, uint4& q2, uint4& q3, bool& pre, const LmRayQ& rq, float tmin, float hitT,
                                            const LmStack& stack, int& sp, const lm_lds_u4* top, uint32_t* boxes = nullptr)
{
    if (boxes) *boxes += ((int)q0.w != LM_REF_NONE) + ((int)q1.w != LM_REF_NONE) + ((int)q2.w != LM_REF_NONE) + ((int)q3.w != LM_REF_NONE);   // counting build
    uint32_t k0, k1, k2, k3;
    lm_slab(q0, rq, tmin, hitT, k0); lm_slab(q1, rq, tmin, hitT, k1); lm_slab(q2, rq, tmin, hitT, k2); lm_slab(q3, rq, tmin, hitT, k3);
    int r0 = (int)q0.w, r1 = (int)q1.w, r2 = (int)q2.w, r3 = (int)q3.w;
    pre = false;
    if (!ANY || LM_ANY_ORDERED) {
        lm_cex(k0, r0, k1, r1); lm_cex(k2, r2, k3, r3); lm_cex(k0, r0, k2, r2);          // (k0, r0) is the nearest hit child now
        if (k0 == 0xffffffffu) return sp == 0 ? LM_REF_NONE : lm_pop(stack, sp);
        pre = r0 >= 0;
        if (pre) lm_node_fetch(sc, r0, top, q0, q1, q2, q3);
        lm_cex(k1, r1, k3, r3); lm_cex(k1, r1, k2, r2);
        if (k3 != 0xffffffffu) lm_push(stack, sp, r3);
        if (k2 != 0xffffffffu) lm_push(stack, sp, r2);
        if (k1 != 0xffffffffu) lm_push(stack, sp, r1);
        return r0;
    }
    // any hit: continue with the first hit child in node order, push the others (highest index first: the same stack as lm_node_step)
    const bool h0 = k0 != 0xffffffffu, h1 = k1 != 0xffffffffu, h2 = k2 != 0xffffffffu, h3 = k3 != 0xffffffffu;
    const int next = h0 ? r0 : h1 ? r1 : h2 ? r2 : h3 ? r3 : LM_REF_NONE;
    if (next == LM_REF_NONE) return sp == 0 ? LM_REF_NONE : lm_pop(stack, sp);
    pre = next >= 0;
    if (pre) lm_node_fetch(sc, next, top, q0, q1, q2, q3);
    if (h3 && (h0 || h1 || h2)) lm_push(stack, sp, r3);
    if (h2 && (h0 || h1)) lm_push(stack, sp, r2);
    if (h1 && h0) lm_push(stack, sp, r1);
    return next;
}
#elif LM_EXPERIMENTS_PART == 3
// Block-aggregated append that also ORDERS the block's appended rays by a 3-bit key (the direction octant), so that the 64 consecutive queue slots a traversal
// wavefront takes hold one or two octants of rays from neighbouring pixels instead of all eight mixed (coherence created where the queue is written,
// no sort pass).  Same single atomic per block; the per-key, per-wave counts are scanned by the first 32 lanes.  `s_key` = 8 x (waves per block) + 1 words, waves <= 4.
// Compile-time switch LM_APPEND_OCTANT (A/B: profiles/r05_append_octant_ab.txt).
__device__ __forceinline__ uint32_t lm_append_slot_block_keyed(uint32_t* counter, bool pred, uint32_t key, uint32_t* s_key)
{
    const uint32_t lane = lm_lane(), wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t mine = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) {
        const unsigned long long m = __ballot(pred && key == k);
        if (lane == 0) s_key[k * nWaves + wave] = (uint32_t)__popcll(m);
        if (key == k) mine = (uint32_t)__popcll(m & below);
    }
    __syncthreads();
    if (threadIdx.x < 64u) {                                  // exclusive scan of the 8 x nWaves (<= 32) counts in key-major order by wave 0
        const uint32_t cnt = threadIdx.x < 8u * nWaves ? s_key[threadIdx.x] : 0u;
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, o, 64); if ((int)lane >= o) inc += up; }
        const uint32_t total = (uint32_t)__shfl((int)inc, 31, 64);
        if (threadIdx.x < 8u * nWaves) s_key[threadIdx.x] = inc - cnt;
        if (threadIdx.x == 0) s_key[8u * nWaves] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = s_key[8u * nWaves] + s_key[key * nWaves + wave] + mine;
    __syncthreads();
    return slot;
}
#elif LM_EXPERIMENTS_PART == 4
// REPACK variant (tuning key tail_repack): a block takes 256 paths, one per lane, and after every depth the surviving paths are packed
// into the block's lowest lanes through LDS (40 bytes of path state: origin, direction, contribution, pixel), so that a wavefront is either full or has no path
// at all — it then skips the depth and only meets the barriers.  Same device functions, RNG streams and per-pixel order of the INDIRECT adds (a pixel has one
// path; its adds are separated by the block barriers): identical image and counters.  Cost: three block barriers per depth, i.e. a depth takes as long as the
// block's slowest wavefront (the plain variant lets every wavefront run ahead on its own).  A/B: profiles/r04_tail_repack_ab.txt.
template <class NEE>
__device__ __forceinline__ void lm_path_tail_repack_body(const LmScene& sc, const LmFrame& fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    __shared__ float s_lut[256];
    __shared__ uint4 s_tab[LM_TABLE_QUADS];
    __shared__ uint32_t s_pack[10 * LM_BLOCK];                   // [word][slot]: consecutive lanes touch consecutive banks
    __shared__ uint32_t s_cnt[LM_BLOCK / 64];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const LmTables tab = lm_stage_tables(s_tab, sc);
    const LmStack stack = lm_make_stack(s_stack, sc);
    const uint32_t n = *inCount;
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = lm_lane(), wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * LM_BLOCK; base < n; base += gridDim.x * LM_BLOCK) {     // block-uniform
        const uint32_t i = base + threadIdx.x;
        bool alive = i < n;
        lf3 o = v3(0.f), d = v3(0.f), c = v3(0.f);
        uint32_t li = 0u;
        if (alive) {
            const float4 o4 = fr.rayO[inQ][i], d4 = fr.rayD[inQ][i], c4 = fr.rayC[inQ][i];
            o = v3(o4); d = v3(d4); c = v3(c4); li = f2u(d4.w);
        }
        uint32_t seed = seed0;
        uint32_t live = min(n - base, (uint32_t)LM_BLOCK);        // paths the block still carries (block-uniform); they sit in threads [0, live)
        for (int depth = depth0; depth < depthMax && live != 0u; depth++) {
            const uint32_t seed2 = lm_wang_hash(seed);
            bool emitRay = false;
            lf3 o2 = v3(0.f), d2 = v3(0.f), c2 = v3(0.f);
            if (wave * 64u < live) {                              // wave-uniform: this wavefront holds paths
                if (depth > depth0) lm_count(fr.counters + LM_CNT_RAYS(depth), alive);
                bool emitShadow = false;
                lf3 sdir = v3(0.f), srad = v3(0.f), spos = v3(0.f);
                float stmax = 0.f;
                if (alive) {
                    LmHit h; h.t = -1.f; h.u = 0.f; h.v = 0.f; h.slot = 0;
                    const bool found = lm_traverse<false>(sc, o, d, 0.01f, 5000.f, stack, h, fr.counters);
                    uint4 rec = make_uint4(0u, 0u, 0u, f2u(-1.f));
                    if (found) {
                        const uint2 id = sc.triId[h.slot];
                        rec = make_uint4(id.x, id.y, lm_f32_to_f16(h.u) | (lm_f32_to_f16(h.v) << 16), f2u(h.t));
                    }
                    LmSurface s;
                    lm_extract(sc, lut, tab, rec, o, d, c, s);
                    const uint32_t ly = li / fr.ww, lx = li - ly * fr.ww;
                    const uint32_t gi = (fr.y0 + ly) * fr.W + (fr.x0 + lx);
                    emitShadow = lm_shade_direct<NEE>(sc, s, gi, seed, sdir, stmax, srad);
                    spos = s.position;
                    if (depth < depthMax - 1) emitRay = lm_shade_indirect(s, gi, seed2, o2, d2, c2);
                }
                lm_count(fr.counters + LM_CNT_SHADOW(depth), emitShadow);
                if (emitShadow) {
                    LmHit hs;
                    if (!lm_traverse<true>(sc, spos, sdir, 0.01f, stmax, stack, hs, fr.counters)) {
                        float4 px = fr.indirect[li];
                        px.x += srad.x; px.y += srad.y; px.z += srad.z;
                        fr.indirect[li] = px;
                    }
                }
            }
            // pack the continuing paths into the lowest threads of the block
            const unsigned long long mask = __ballot(emitRay);
            if (lane == 0u) s_cnt[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0u, total = 0u;
#pragma unroll
            for (uint32_t w = 0; w < LM_BLOCK / 64u; w++) { const uint32_t k = s_cnt[w]; before += w < wave ? k : 0u; total += k; }
            if (emitRay) {
                const uint32_t slot = before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                s_pack[slot] = f2u(o2.x); s_pack[LM_BLOCK + slot] = f2u(o2.y); s_pack[2 * LM_BLOCK + slot] = f2u(o2.z);
                s_pack[3 * LM_BLOCK + slot] = f2u(d2.x); s_pack[4 * LM_BLOCK + slot] = f2u(d2.y); s_pack[5 * LM_BLOCK + slot] = f2u(d2.z);
                s_pack[6 * LM_BLOCK + slot] = f2u(c2.x); s_pack[7 * LM_BLOCK + slot] = f2u(c2.y); s_pack[8 * LM_BLOCK + slot] = f2u(c2.z);
                s_pack[9 * LM_BLOCK + slot] = li;
            }
            __syncthreads();
            alive = threadIdx.x < total;
            if (alive) {
                const uint32_t t = threadIdx.x;
                o = v3(u2f(s_pack[t]), u2f(s_pack[LM_BLOCK + t]), u2f(s_pack[2 * LM_BLOCK + t]));
                d = v3(u2f(s_pack[3 * LM_BLOCK + t]), u2f(s_pack[4 * LM_BLOCK + t]), u2f(s_pack[5 * LM_BLOCK + t]));
                c = v3(u2f(s_pack[6 * LM_BLOCK + t]), u2f(s_pack[7 * LM_BLOCK + t]), u2f(s_pack[8 * LM_BLOCK + t]));
                li = s_pack[9 * LM_BLOCK + t];
            }
            live = total;
            __syncthreads();                                      // s_cnt / s_pack are rewritten in the next depth (and the INDIRECT adds of this depth are visible to it)
            seed = lm_wang_hash(seed);
        }
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail_repack)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0)
{ lm_path_tail_repack_body<LmExact>(sc, fr, inQ, inCount, depth0, depthMax, seed0); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail_repack_fs)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0)
{ lm_path_tail_repack_body<LmFast>(sc, fr, inQ, inCount, depth0, depthMax, seed0); }
#endif
#undef LM_EXPERIMENTS_PART
