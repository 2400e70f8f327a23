// kernels.hip — hand-written HIP kernels (gfx950) of the wavefront path tracer.
//
// One frame = primary-ray generation, then per wave: closest-hit traversal of the 4-wide BVH, surface extraction + shading
// (ReSTIR DI at depth 0, CDF next-event estimation afterwards), Russian-roulette continuation with wavefront
// ballot/prefix-sum compaction, any-hit shadow rays, and finally channel merge + output.  Behaviour follows the
// reference kernels cited at each function (paths relative to /root/reference/Lumen_Engine/LumenPT/src);
// structure does not: queues are SoA float4 streams, depth>=1 extraction+NEE+continuation are one fused
// kernel, shadow rays are resolved per wave (one fp32 add per pixel per wave), launches are fixed-size
// grid-stride over device-side counters so a frame needs no host round trip.
#include "lm_layout.h"
#include "lm_bsdf.h"
#include "lm_tri.h"

#define LM_BLOCK 256
#ifndef LM_RESTIR_WAVES
#define LM_RESTIR_WAVES 1        // __launch_bounds__ minimum waves per SIMD for the ALU-heavy ReSTIR kernels (tuning knob)
#endif
#ifndef LM_FAST_WAVES
#define LM_FAST_WAVES 5          // the fast-mode instantiations of those kernels fit 96 VGPRs: five waves per SIMD
#endif
#ifndef LM_INSTRUMENT
#define LM_INSTRUMENT 0
#endif
#ifndef LM_SHADE_PRIO
#define LM_SHADE_PRIO 3          // s_setprio of the surface-extraction / shading kernels of the wave chain: with the fast ReSTIR mode that chain is the
#endif                           // critical path, and its VALU-heavy kernels otherwise queue behind the candidate pick on every SIMD (+1.5 % on C2)
// the file is compiled twice into one library: kernel symbols of the counting build get a suffix
// LM_NOSLP_VARIANT: the same kernels once more under the names *_ns, from a compilation with -fno-slp-vectorize (Makefile: kernels_ns.o).  The compiler's packed-fp32 pairing
// buys no issue slots on this chip and costs moves and register pairs; which kernels run from which compilation is a run-time choice of the renderer (renderer.cpp
// `noslp_kernels`, environment LUMEN_MI_NOSLP_KERNELS), so that one build can be A/B-ed kernel by kernel on one box (profiles/r06_noslp_kernels_ab.txt).
#ifndef LM_NOSLP_VARIANT
#define LM_NOSLP_VARIANT 0
#endif
#if LM_INSTRUMENT
#define KN(x) x##_inst
#elif LM_NOSLP_VARIANT
#define KN(x) x##_ns
#else
#define KN(x) x
#endif

// ---------------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lm_lane() { return __lane_id(); }

// wavefront-aggregated append: one atomic per 64-lane wave (ballot + prefix popcount)
__device__ __forceinline__ uint32_t lm_append_slot(uint32_t* counter, bool pred)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t lane = lm_lane();
    const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + prefix;
}
__device__ __forceinline__ void lm_count(uint32_t* counter, bool pred)
{
    const unsigned long long mask = __ballot(pred);
    if (mask != 0ull && (int)lm_lane() == __ffsll((long long)mask) - 1) atomicAdd(counter, (uint32_t)__popcll(mask));
}
// Block-aggregated forms.  One returning atomic on ONE address retires at only ~88 per microsecond on MI355X, so a launch
// must not issue one atomic per wavefront on a shared counter (57 600 waves at 1440p = 0.65 ms of pure atomic time):
// the waves of a block first combine through LDS and the block issues a single atomic.  Every thread of the block must call
// these (they contain barriers).  `s_tmp` = LDS scratch of (waves per block + 1) words.
__device__ __forceinline__ uint32_t lm_append_slot_block(uint32_t* counter, bool pred, uint32_t* s_tmp)
{
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = lm_lane(), wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_tmp[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < nWaves; w++) { const uint32_t c = s_tmp[w]; s_tmp[w] = total; total += c; }
        s_tmp[nWaves] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = s_tmp[nWaves] + s_tmp[wave] + prefix;
    __syncthreads();                      // s_tmp may be reused by the caller's next iteration
    return slot;
}
// two appends behind one set of barriers (s_tmp: 2 x (waves per block + 1) words)
__device__ __forceinline__ void lm_append_slot_block2(uint32_t* counterA, bool predA, uint32_t& slotA, uint32_t* counterB, bool predB, uint32_t& slotB, uint32_t* s_tmp)
{
    const unsigned long long maskA = __ballot(predA), maskB = __ballot(predB);
    const uint32_t lane = lm_lane(), wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t* tA = s_tmp; uint32_t* tB = s_tmp + nWaves + 1u;
    if (lane == 0) { tA[wave] = (uint32_t)__popcll(maskA); tB[wave] = (uint32_t)__popcll(maskB); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t totalA = 0, totalB = 0;
        for (uint32_t w = 0; w < nWaves; w++) { const uint32_t a = tA[w]; tA[w] = totalA; totalA += a; const uint32_t b = tB[w]; tB[w] = totalB; totalB += b; }
        tA[nWaves] = totalA ? atomicAdd(counterA, totalA) : 0u;
        tB[nWaves] = totalB ? atomicAdd(counterB, totalB) : 0u;
    }
    __syncthreads();
    slotA = tA[nWaves] + tA[wave] + (uint32_t)__popcll(maskA & below);
    slotB = tB[nWaves] + tB[wave] + (uint32_t)__popcll(maskB & below);
    __syncthreads();
}
#ifndef LM_APPEND_OCTANT
#define LM_APPEND_OCTANT 0
#endif
__device__ __forceinline__ uint32_t lm_octant(const lf3& d) { return (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u); }
#if LM_APPEND_OCTANT
#define LM_EXPERIMENTS_PART 3      // octant-ordered block append (round 5: measured, -0.4 %)
#include "lm_experiments.h"
#endif
__device__ __forceinline__ void lm_count_block(uint32_t* counter, bool pred, uint32_t* s_tmp)
{
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = lm_lane(), wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    if (lane == 0) s_tmp[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < nWaves; w++) total += s_tmp[w];
        if (total) atomicAdd(counter, total);
    }
    __syncthreads();
}

__device__ __forceinline__ float4 lm_mul_m34(const float* m, const lf3& v, float w)   // rows 0..2 of sutil Matrix4x4 * float4
{
    float4 r;
    r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w;
    r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w;
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w;
    r.w = 0.f;
    return r;
}

#include "lm_traverse.h"

// ---------------------------------------------------------------------------------------------------------------------
// K1: primary rays — reference GPUGeneratePrimRay.cu:28-82 (Halton(2,3) jitter indexed by frameCount + pixel index).
// Ray slot i holds the pixel of an 8x8-tile enumeration of the window (every consumer finds the pixel in rayD.w), so one
// wavefront traces an 8x8 pixel bundle instead of a 64x1 strip.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lm_slot_to_pixel(const LmFrame& fr, uint32_t i, uint32_t& lx, uint32_t& ly)
{
    const uint32_t w8 = fr.ww & ~7u, h8 = fr.wh & ~7u, nA = w8 * h8;
    if (i < nA) {
        const uint32_t t = i >> 6, l = i & 63u, tpr = w8 >> 3;
        lx = (t % tpr) * 8u + (l & 7u); ly = (t / tpr) * 8u + (l >> 3);
        return;
    }
    uint32_t j = i - nA;
    const uint32_t rw = fr.ww - w8;
    if (j < rw * h8) { ly = j / rw; lx = w8 + j % rw; return; }
    j -= rw * h8;
    ly = h8 + j / fr.ww; lx = j % fr.ww;
}
// primary ray of queue slot i: its pixel (window-local index) and direction — GPUGeneratePrimRay.cu:28-82 (Halton jitter of index frameCount + pixel)
__device__ __forceinline__ lf3 lm_primary_dir(const LmFrame& fr, const LmCamera& cam, uint32_t frameCount, uint32_t i, uint32_t& li)
{
    uint32_t lx, ly;
    lm_slot_to_pixel(fr, i, lx, ly);
    li = ly * fr.ww + lx;
    const uint32_t sx = fr.x0 + lx, sy = fr.y0 + ly;
    const uint32_t gi = sy * fr.W + sx;
    const float jx = lm_halton(frameCount + gi, 2u), jy = lm_halton(frameCount + gi, 3u);
    float dx = ((float)(int)sx + jx) / (float)fr.W;
    float dy = ((float)(int)sy + jy) / (float)fr.H;
    dx = -(dx * 2.0f - 1.0f);
    dy = -(dy * 2.0f - 1.0f);
    const lf3 U = v3(cam.U[0], cam.U[1], cam.U[2]), V = v3(cam.V[0], cam.V[1], cam.V[2]), Wv = v3(cam.Wv[0], cam.Wv[1], cam.Wv[2]);
    return normalize3(dx * U + dy * V + Wv);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_primary)(LmFrame fr, LmCamera cam, uint32_t frameCount)
{
#if LM_PRIMARY_CLEARS
    if (blockIdx.x == 0) {              // the frame's counter block (this frame parity's; every other kernel of the frame runs behind this one)
        for (uint32_t w = threadIdx.x; w < LM_CNT_WORDS; w += LM_BLOCK) fr.counters[w] = 0u;
        __syncthreads();
    }
#endif
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < fr.n; i += stride) {
        uint32_t li;
        const lf3 dir = lm_primary_dir(fr, cam, frameCount, i, li);
        // only the direction plane is written: every primary ray starts at the eye with contribution (1, 1, 1), which the first closest-hit
        // launch and the depth-0 extraction take from their arguments instead of reading 32 bytes per pixel back (IntersectionRayData's
        // origin and contribution, GPUGeneratePrimRay.cu:69-72)
        fr.rayD[0][i] = make_float4(dir.x, dir.y, dir.z, u2f(li));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) fr.counters[LM_CNT_RAYS(0)] = fr.n;
}

// ---------------------------------------------------------------------------------------------------------------------
// K2-K4: closest-hit query — reference WaveFrontShaders.cu:42-76,301-340 (tmin 0.01, tmax 5000, miss => t = -1)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef LM_TRACE_WAVES
#define LM_TRACE_WAVES 8      // <= 64 VGPRs: eight waves per SIMD
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_TRACE_WAVES)
KN(lm_k_trace_closest)(LmScene sc, const float4* __restrict__ rayO /* NULL: every ray starts at `eye` (primary rays) */, const float4* __restrict__ rayD,
                   const uint32_t* __restrict__ countPtr, uint4* __restrict__ hits, float tmin, float tmax, uint32_t* counters, int refillBelow, float4 eye)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    const uint32_t n = *countPtr;
    lm_trace_queue<false>(sc, n, refillBelow, lm_make_stack(s_stack, sc), lm_stage_top(s_top, sc), counters,
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) { o = rayO ? v3(rayO[i]) : v3(eye); d = v3(rayD[i]); t0 = tmin; t1 = tmax; },
        [&](uint32_t i, bool found, const LmHit& h) {
            uint4 out = make_uint4(0u, 0u, 0u, f2u(-1.f));
            if (found) {
                const uint2 id = sc.triId[h.slot];
                out.x = id.x; out.y = id.y;
                out.z = lm_f32_to_f16(h.u) | (lm_f32_to_f16(h.v) << 16);      // half2 barycentrics (IntersectionData.h:90)
                out.w = f2u(h.t);
            }
            hits[i] = out;
        });
}

// the same query for a queue of COHERENT rays (the primary wave): packet traversal, one shared stack per wavefront (lm_traverse.h)
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_TRACE_WAVES)
KN(lm_k_trace_closest_packet)(LmScene sc, const float4* __restrict__ rayO, const float4* __restrict__ rayD, const uint32_t* __restrict__ countPtr,
                          uint4* __restrict__ hits, float tmin, float tmax, float4 eye)
{
    __shared__ int s_wstack[LM_PACKET_STACK * (LM_BLOCK / 64)];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    const uint32_t n = *countPtr;
    lm_trace_packets<false>(sc, n, (lm_lds_int*)(s_wstack + LM_PACKET_STACK * (threadIdx.x >> 6)), lm_stage_top(s_top, sc),
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) { o = rayO ? v3(rayO[i]) : v3(eye); d = v3(rayD[i]); t0 = tmin; t1 = tmax; },
        [&](uint32_t i, bool found, const LmHit& h) {
            uint4 out = make_uint4(0u, 0u, 0u, f2u(-1.f));
            if (found) {
                const uint2 id = sc.triId[h.slot];
                out.x = id.x; out.y = id.y;
                out.z = lm_f32_to_f16(h.u) | (lm_f32_to_f16(h.v) << 16);
                out.w = f2u(h.t);
            }
            hits[i] = out;
        });
}

#include "lm_shade.h"

// K7 (depth 0) + K9 motion vectors (MotionVectors.cu:8-55) + K10 ResolveDirectLightHits (GPUShadeDirect.cu:11-40) + channel clear
// + K12 at depth 0 (GPUShadeIndirect.cu:7-146): the path continuation is sampled from the surface while it is still in
// registers; survivors of a block iteration are appended to the wave-1 queue with ONE atomic.
#ifndef LM_EXTRACT_WAVES
#define LM_EXTRACT_WAVES 5       // minimum waves per SIMD asked of the compiler for the depth-0 kernel / the wave shading kernel.  Round 6: 5 / 5 — the instantiations that run
#endif                           // (from the compilation without the SLP vectoriser: 104 / 112 registers unbounded) fit 96 registers with 0 / 11 scratch instructions: five waves instead of
#ifndef LM_SHADE_WAVES           // four, +1.1 % fast eager, +1.2 % exact (profiles/r06_extract_waves_ab.txt).  Rounds 1 - 5: 1 = no bound (120 / 124 registers with the SLP pass: the bound spilled).
#define LM_SHADE_WAVES 5
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_EXTRACT_WAVES)
KN(lm_k_extract0)(LmScene sc, LmFrame fr, LmCamera cam, int cur, uint32_t seed2, int doIndirect, int outQ, uint32_t* outCount)
{
    __shared__ uint32_t s_tmp[10];
#if LM_APPEND_OCTANT
    __shared__ uint32_t s_key[8 * (LM_BLOCK / 64) + 1];
#endif
    __shared__ float s_lut[256];
    __shared__ uint4 s_tab[LM_TABLE_QUADS];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const LmTables tab = lm_stage_tables(s_tab, sc);
#if LM_SHADE_PRIO
    __builtin_amdgcn_s_setprio(LM_SHADE_PRIO);
#endif
    const uint32_t stride = gridDim.x * LM_BLOCK;
    const uint32_t nIter = (fr.n + stride - 1u) / stride;
    for (uint32_t it = 0; it < nIter; it++) {
        const uint32_t i = it * stride + blockIdx.x * LM_BLOCK + threadIdx.x;
        bool emit = false, keeps = false;
        uint32_t liKeeps = 0u;
        lf3 bo = v3(0.f), bd = v3(0.f), bc = v3(0.f);
        uint32_t liOut = 0u;
        if (i < fr.n) {
        const float4 d4 = fr.rayD[0][i];                          // origin = eye, contribution = 1 (lm_k_primary)
        const uint32_t li = f2u(d4.w);
        LmSurface s;
        lm_extract(sc, lut, tab, fr.hits[i], v3(cam.eye[0], cam.eye[1], cam.eye[2]), v3(d4), v3(1.f, 1.f, 1.f), s);
        lm_gbuf_store(fr.gbuf[cur], fr.probe[cur], li, s);
        // does any surface of the frame need the second (exact) launch of the fast ReSTIR passes?  A FLAG, not a count: one plain store per
        // wavefront that sees such a surface (an atomic per wavefront on one address is ~ 88 per microsecond: 0.65 ms at 1440p in a scene of
        // glass or clear coat, on the critical wave chain, in every mode)
        const bool rareSurface = !s.flags && !lm_quick_contracts(s.mat);
        if (__ballot(rareSurface) != 0ull && lm_lane() == 0u) fr.counters[LM_CNT_RARE] = 1u;
        if (rareSurface && fr.rareTile[cur]) {                      // ... and WHERE: the pixel's 16 x 16 tile of the window (lm_rare_near)
            const uint32_t ry = li / fr.ww, rx = li - ry * fr.ww;
            fr.rareTile[cur][(ry >> 4) * ((fr.ww + 15u) >> 4) + (rx >> 4)] = 1u;
        }
        // lazy reuse: the previous frame left its history passes pending.  Pixels that were reuse surfaces then and are flagged now keep their reservoir entry
        // past this frame's candidate pick: listed for lm_k_reuse_counts (silhouette pixels under sub-pixel jitter; more when the camera moves)
        // (appended below, with the continuation rays: one atomic per block, not per wavefront — a camera cut can flag every pixel)
        if (fr.owedSet >= 0) { keeps = s.flags != 0u && fr.probe[fr.owedSet][li].w >= 0.f; liKeeps = li; }
        // motion vector
        const uint32_t ly = li / fr.ww, lx = li - ly * fr.ww;
        const uint32_t px = fr.x0 + lx, py = fr.y0 + ly;
        uint32_t mv = 0u;
        if (s.t > 0.f) {
            float csx = (float)(int)px, csy = (float)(int)py;
            csx += 0.5f; csy += 0.5f;
            csx /= (float)fr.W; csy /= (float)fr.H;
            const float* M = cam.prevViewProj;
            const float cx = M[0] * s.position.x + M[1] * s.position.y + M[2] * s.position.z + M[3] * 1.0f;
            const float cy = M[4] * s.position.x + M[5] * s.position.y + M[6] * s.position.z + M[7] * 1.0f;
            const float cz = M[8] * s.position.x + M[9] * s.position.y + M[10] * s.position.z + M[11] * 1.0f;
            const float cw = M[12] * s.position.x + M[13] * s.position.y + M[14] * s.position.z + M[15] * 1.0f;
            const lf3 ndc = v3(cx, cy, cz) / cw;
            const float psx = ndc.x * 0.5f + 0.5f, psy = ndc.y * 0.5f + 0.5f;
            mv = lm_f32_to_f16(psx - csx) | (lm_f32_to_f16(psy - csy) << 16);
        }
        fr.motion[li] = mv;
        fr.direct[li] = (s.flags & LM_SF_EMISSIVE) ? s.mat.color : make_float4(0.f, 0.f, 0.f, 0.f);
        fr.indirect[li] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (doIndirect && lm_owned(fr, li, 0)) {                  // halo pixels need no indirect light: their radiance is discarded
            s.transport = v3(1.f, 1.f, 1.f);                       // what a G-buffer reload yields (lm_gbuf_load)
            liOut = li;
            emit = lm_shade_indirect(s, py * fr.W + px, seed2, bo, bd, bc);
        }
        }
        if (doIndirect || fr.owedSet >= 0) {                       // uniform per block
            uint32_t slot, slotKeeps;
#if LM_APPEND_OCTANT
            slot = lm_append_slot_block_keyed(outCount, emit, lm_octant(bd), s_key);
            slotKeeps = fr.owedSet >= 0 ? lm_append_slot_block(fr.counters + LM_CNT_HAZARD, keeps, s_tmp) : 0u;
#else
            lm_append_slot_block2(outCount, emit, slot, fr.counters + LM_CNT_HAZARD, keeps, slotKeeps, s_tmp);
#endif
            if (emit) {
                fr.rayO[outQ][slot] = v4(bo, 0.f);
                fr.rayD[outQ][slot] = v4(bd, u2f(liOut));
                fr.rayC[outQ][slot] = v4(bc, 0.f);
            }
            if (keeps) fr.hazardList[slotKeeps] = liKeeps;
        }
    }
}


// depth >= 1: extraction + NEE + continuation fused (no SurfaceData round trip through HBM)
template <class NEE>
__device__ __forceinline__ void lm_shade_wave_body(const LmScene& sc, const LmFrame& fr, int inQ, const uint32_t* __restrict__ inCount, uint32_t seed, uint32_t seed2, int doIndirect,
                uint32_t* outCount, uint32_t* shadowCount)
{
    __shared__ uint32_t s_tmp[5];
#if LM_APPEND_OCTANT
    __shared__ uint32_t s_key[8 * (LM_BLOCK / 64) + 1];
#endif
    __shared__ float s_lut[256];
    __shared__ uint4 s_tab[LM_TABLE_QUADS];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const LmTables tab = lm_stage_tables(s_tab, sc);
#if LM_SHADE_PRIO
    __builtin_amdgcn_s_setprio(LM_SHADE_PRIO);
#endif
    const uint32_t n = *inCount;
    const uint32_t stride = gridDim.x * LM_BLOCK;
    const uint32_t nIter = (n + stride - 1u) / stride;
    const int outQ = inQ ^ 1;
    for (uint32_t it = 0; it < nIter; it++) {
        const uint32_t i = it * stride + blockIdx.x * LM_BLOCK + threadIdx.x;
        bool emitShadow = false, emitRay = false;
        lf3 sdir = v3(0.f), srad = v3(0.f), o = v3(0.f), d = v3(0.f), c = v3(0.f), spos = v3(0.f);
        float stmax = 0.f;
        uint32_t li = 0;
        if (i < n) {
            const float4 o4 = fr.rayO[inQ][i], d4 = fr.rayD[inQ][i], c4 = fr.rayC[inQ][i];
            li = f2u(d4.w);
            LmSurface s;
            lm_extract(sc, lut, tab, fr.hits[i], v3(o4), v3(d4), v3(c4), s);
            const uint32_t ly = li / fr.ww, lx = li - ly * fr.ww;
            const uint32_t gi = (fr.y0 + ly) * fr.W + (fr.x0 + lx);
            emitShadow = lm_shade_direct<NEE>(sc, s, gi, seed, sdir, stmax, srad);
            spos = s.position;
            if (doIndirect) emitRay = lm_shade_indirect(s, gi, seed2, o, d, c);
        }
#if LM_APPEND_OCTANT
        const uint32_t ss = lm_append_slot_block_keyed(shadowCount, emitShadow, lm_octant(sdir), s_key);
#else
        const uint32_t ss = lm_append_slot_block(shadowCount, emitShadow, s_tmp);
#endif
        if (emitShadow) {
            fr.shO[ss] = v4(spos, stmax);
            fr.shD[ss] = v4(sdir, u2f(li));
            fr.shR[ss] = v4(srad, 0.f);
        }
#if LM_APPEND_OCTANT
        const uint32_t rs = lm_append_slot_block_keyed(outCount, emitRay, lm_octant(d), s_key);
#else
        const uint32_t rs = lm_append_slot_block(outCount, emitRay, s_tmp);
#endif
        if (emitRay) {
            fr.rayO[outQ][rs] = v4(o, 0.f);
            fr.rayD[outQ][rs] = v4(d, u2f(li));
            fr.rayC[outQ][rs] = v4(c, 0.f);
        }
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SHADE_WAVES)
KN(lm_k_shade_wave)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, uint32_t seed, uint32_t seed2, int doIndirect, uint32_t* outCount, uint32_t* shadowCount)
{ lm_shade_wave_body<LmExact>(sc, fr, inQ, inCount, seed, seed2, doIndirect, outCount, shadowCount); }
// tuning key fast_shade: the NEE contribution with hardware reciprocal / square root (lm_shade.h lm_shade_direct)
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_shade_wave_fs)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, uint32_t seed, uint32_t seed2, int doIndirect, uint32_t* outCount, uint32_t* shadowCount)
{ lm_shade_wave_body<LmFast>(sc, fr, inQ, inCount, seed, seed2, doIndirect, outCount, shadowCount); }

// Path tail: waves `depth0 .. depthMax-1` of the rays left in the queue, one path per lane, in ONE launch (closest hit ->
// extraction + NEE + continuation -> shadow ray -> next depth).  Deep waves hold too few rays to fill the machine, so a
// wave-per-launch schedule pays, per depth, the dependent chain of the longest ray of the whole queue (K2-K5 + K7 + K11 +
// K12 three launches per depth).  Here a wavefront only waits for its own `lanesPerWave` paths.  Arithmetic, RNG streams
// and the order of the INDIRECT adds per pixel are those of the per-wave kernels (same device functions), so the result
// is identical; ray counters are accumulated per depth like the queue appends do.
template <bool PAIR, class NEE>
__device__ __forceinline__ void lm_path_tail_body(const LmScene& sc, const LmFrame& fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    __shared__ float s_lut[256];
    __shared__ uint4 s_tab[LM_TABLE_QUADS];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const LmTables tab = lm_stage_tables(s_tab, sc);
    const LmStack stack = lm_make_stack(s_stack, sc);
    const uint32_t n = *inCount;
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = lm_lane(), L = (uint32_t)(lanesPerWave < 0 ? -lanesPerWave : lanesPerWave);
    const uint32_t W = gridDim.x * (LM_BLOCK / 64u);
    if constexpr (PAIR) {
        // PAIR MODE (L <= 32): lanes [0, L) carry the paths, lanes [L, 2L) their partners.  The NEE shadow ray a path emits at depth d is handed
        // to its partner lane and traced in the NEXT round, beside the path's own closest-hit query of depth d + 1: a depth costs
        // max(closest hit, shadow ray) + shading instead of their sum.  Per pixel the INDIRECT adds still happen in depth order (the add of
        // depth d is made in round d + 1, before that round's shading can emit the next shadow ray), with the same operands: identical image.
        const bool isPath = lane < L, isShadow = lane >= L && lane < 2u * L;
        const int partner = (int)lane - (int)L;                      // the path lane a shadow lane serves
        for (uint32_t base = (blockIdx.x * (LM_BLOCK / 64u) + (threadIdx.x >> 6)) * L; base < n; base += W * L) {     // wave-uniform
            const uint32_t i = base + lane;
            bool alive = isPath && i < n;
            lf3 o = v3(0.f), d = v3(0.f), c = v3(0.f);
            uint32_t li = 0u;
            if (alive) {
                const float4 o4 = fr.rayO[inQ][i], d4 = fr.rayD[inQ][i], c4 = fr.rayC[inQ][i];
                o = v3(o4); d = v3(d4); c = v3(c4); li = f2u(d4.w);
            }
            bool shValid = false;                                     // shadow lanes: a ray is waiting
            lf3 shO = v3(0.f), shD = v3(0.f), shR = v3(0.f);
            float shT = 0.f;
            uint32_t shLi = 0u;
            uint32_t seed = seed0;
            for (int depth = depth0; depth <= depthMax; depth++) {     // one extra round resolves the last depth's shadow rays
                const bool tracePath = alive && depth < depthMax;
                if (depth > depth0 && depth < depthMax) lm_count(fr.counters + LM_CNT_RAYS(depth), alive);
                if (__ballot(tracePath || shValid) == 0ull) break;
                const uint32_t seed2 = lm_wang_hash(seed);
                LmHit h; h.t = -1.f; h.u = 0.f; h.v = 0.f; h.slot = 0;
                const bool found = lm_traverse_mixed(sc, isShadow ? shO : o, isShadow ? shD : d, 0.01f, isShadow ? shT : 5000.f, isShadow, tracePath || shValid, stack, h, fr.counters);
                if (shValid && !found) {
                    float4 px = fr.indirect[shLi];
                    px.x += shR.x; px.y += shR.y; px.z += shR.z;
                    fr.indirect[shLi] = px;
                }
                shValid = false;
                bool emitShadow = false, emitRay = false;
                lf3 sdir = v3(0.f), srad = v3(0.f), spos = v3(0.f), o2 = v3(0.f), d2 = v3(0.f), c2 = v3(0.f);
                float stmax = 0.f;
                if (tracePath) {
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
                if (depth < depthMax) lm_count(fr.counters + LM_CNT_SHADOW(depth), emitShadow);
                // hand the shadow ray to the partner lane (every lane takes part in the shuffles)
                {
                    const int src = isShadow ? partner : (int)lane;
                    const int ev = __shfl((int)emitShadow, src, 64);
                    const float ax = __shfl(spos.x, src, 64), ay = __shfl(spos.y, src, 64), az = __shfl(spos.z, src, 64);
                    const float bx = __shfl(sdir.x, src, 64), by = __shfl(sdir.y, src, 64), bz = __shfl(sdir.z, src, 64);
                    const float cx = __shfl(srad.x, src, 64), cy = __shfl(srad.y, src, 64), cz = __shfl(srad.z, src, 64);
                    const float tm = __shfl(stmax, src, 64);
                    const uint32_t pl = (uint32_t)__shfl((int)li, src, 64);
                    if (isShadow) { shValid = ev != 0; shO = v3(ax, ay, az); shD = v3(bx, by, bz); shR = v3(cx, cy, cz); shT = tm; shLi = pl; }
                }
                alive = emitRay;
                o = o2; d = d2; c = c2;
                seed = lm_wang_hash(seed);
            }
        }
        return;
    } else {
    for (uint32_t base = (blockIdx.x * (LM_BLOCK / 64u) + (threadIdx.x >> 6)) * L; base < n; base += W * L) {     // wave-uniform
        const uint32_t i = base + lane;
        bool alive = lane < L && i < n;
        lf3 o = v3(0.f), d = v3(0.f), c = v3(0.f);
        uint32_t li = 0u;
        if (alive) {
            const float4 o4 = fr.rayO[inQ][i], d4 = fr.rayD[inQ][i], c4 = fr.rayC[inQ][i];
            o = v3(o4); d = v3(d4); c = v3(c4); li = f2u(d4.w);
        }
        uint32_t seed = seed0;
        for (int depth = depth0; depth < depthMax; depth++) {
            if (depth > depth0) lm_count(fr.counters + LM_CNT_RAYS(depth), alive);         // LM_CNT_RAYS(depth0) was written by the producer of the queue
            if (__ballot(alive) == 0ull) break;
            const uint32_t seed2 = lm_wang_hash(seed);
            bool emitShadow = false, emitRay = false;
            lf3 sdir = v3(0.f), srad = v3(0.f), spos = v3(0.f), o2 = v3(0.f), d2 = v3(0.f), c2 = v3(0.f);
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
            alive = emitRay;
            o = o2; d = d2; c = c2;
            seed = lm_wang_hash(seed);
        }
    }
    }
}
// Residency of the path tail.  The tail is one long launch beside the frame's critical ReSTIR chain: every wave slot and register it holds is taken from the kernels
// that bound the frame, while its own length is the dependent chain of its longest path.  LM_TAIL_MAX_WAVES caps its waves per SIMD (0: whatever its registers allow);
// interleaved A/B: profiles/r06_build_flags_ab.txt.
#ifndef LM_TAIL_MAX_WAVES
#define LM_TAIL_MAX_WAVES 3
#endif
#if LM_TAIL_MAX_WAVES
#define LM_TAIL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(1, LM_TAIL_MAX_WAVES)))
#else
#define LM_TAIL_OCCUPANCY
#endif
#define LM_EXPERIMENTS_PART 4      // the repacking path tail (round 4: measured, -1.4 %; tuning key tail_repack keeps it reachable)
#include "lm_experiments.h"
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{ lm_path_tail_body<false, LmExact>(sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail_pair)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{ lm_path_tail_body<true, LmExact>(sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail_fs)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{ lm_path_tail_body<false, LmFast>(sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK) LM_TAIL_OCCUPANCY
KN(lm_k_path_tail_pair_fs)(LmScene sc, LmFrame fr, int inQ, const uint32_t* __restrict__ inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{ lm_path_tail_body<true, LmFast>(sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave); }

// K5: NEE shadow rays — reference WaveFrontShaders.cu:114-179 (tmin 0.01; unoccluded => channel += radiance).
// At most one shadow ray per pixel per wave, so the add is a plain fp32 read-modify-write.
#ifndef LM_ANYHIT_WAVES
#define LM_ANYHIT_WAVES 8      // minimum waves per SIMD of the two any-hit queue kernels: capped at 64 VGPRs like the closest-hit kernel (the watertight triangle test took them to 68 / 72 = seven waves; A/B profiles/r06_watertight_ab.txt)
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_ANYHIT_WAVES)
KN(lm_k_trace_shadow)(LmScene sc, LmFrame fr, const uint32_t* __restrict__ countPtr, float tmin, int refillBelow)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    const uint32_t n = *countPtr;
    lm_trace_queue<true>(sc, n, refillBelow, lm_make_stack(s_stack, sc), lm_stage_top(s_top, sc), fr.counters,
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) { const float4 o4 = fr.shO[i]; o = v3(o4); d = v3(fr.shD[i]); t0 = tmin; t1 = o4.w; },
        [&](uint32_t i, bool occluded, const LmHit&) {
            if (!occluded) {
                const uint32_t li = f2u(fr.shD[i].w);
                const float4 r = fr.shR[i];
                float4 px = fr.indirect[li];
                px.x += r.x; px.y += r.y; px.z += r.z;
                fr.indirect[li] = px;
            }
        });
}

// reservoir buffer behind an LM_RES_* code (lm_layout.h)
__device__ __forceinline__ int lm_res_idx(const LmFrame& fr, int code)
{
    if (code >= 0) return code;
    if (code == LM_RES_OWED) return fr.swap[4] & 1;
    const int cur = *fr.swap & 1;
    return code == LM_RES_CUR ? cur : cur ^ 1;
}
// Deferred history passes (frame.cpp, "lazy reuse").  Both spatial passes and the combine of a frame only build what a LATER temporal pass reads as
// "previous"; nothing of the frame itself depends on them.  They are launched at the start of the NEXT frame's ReSTIR chain (fr.deferred != 0) and run
// only if the swap chain has turned since (swap[0] != swap[4], grid-uniform): the buffer they complete is then the history — every frame of an odd path
// depth.  Otherwise (even path depth: the reference's swap quirk) this frame's candidate pick overwrites that buffer before anything reads it —
// except at pixels which are flagged (emitter / miss) now: there the pick only zeroes the weight (ReSTIRKernels.cu:441-447) and the rest of the entry can
// be read frames later through a probe plane of another age.  lm_k_reuse_counts completes exactly those entries.  swap[5] = 1: run already (history
// export between frames) or settled.
__device__ __forceinline__ bool lm_reuse_turned(const LmFrame& fr) { return (fr.swap[4] & 1) != (*fr.swap & 1); }
__device__ __forceinline__ bool lm_reuse_owed(const LmFrame& fr) { return fr.swap[5] == 0 && lm_reuse_turned(fr); }
// does a ReSTIR pass of the fast mode's second (exact) launch have anything to do?  (The deferred passes belong to the previous frame: its flag was
// parked in swap[7] by that frame's merge, its counter block may already belong to the frame after this one.)
// Fast mode, second (exact) launch of a ReSTIR pass: does the window-local pixel rectangle [x0, x1] x [y0, y1] touch a tile in which the extraction of G-buffer set `cur` saw a
// surface outside the contracted evaluation?  Block-uniform (contains a barrier: every thread of the block calls it, all return the same).  Before round 5 every block of
// the second launch loaded each pixel's 128-byte surface record to learn that it had nothing to do: 4 launches x 30 - 40 us per 720p TraceFrame of the reference's default
// model for the 0.08 % of its surfaces that are glass, and ~ 0.4 ms at 1440p for a single such pixel.
__device__ __forceinline__ bool lm_rare_near(const LmFrame& fr, int cur, int x0, int y0, int x1, int y1)
{
#ifndef LM_RARE_TILES
#define LM_RARE_TILES 1            // 0: A/B switch, every block of the second launch looks at its pixels (profiles/r05_rare_tiles_ab.txt)
#endif
    const uint32_t* __restrict__ map = fr.rareTile[cur];
    if (!LM_RARE_TILES || !map) return true;
    const int tilesX = (int)((fr.ww + 15u) >> 4), tilesY = (int)((fr.wh + 15u) >> 4);
    const int tx0 = max(x0, 0) >> 4, ty0 = max(y0, 0) >> 4, tx1 = min(x1 >> 4, tilesX - 1), ty1 = min(y1 >> 4, tilesY - 1);
    const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;
    int any = 0;
    if (x1 >= 0 && y1 >= 0 && nx > 0 && ny > 0)
        for (int k = (int)threadIdx.x; k < nx * ny; k += (int)blockDim.x) any |= (int)map[(ty0 + k / nx) * tilesX + tx0 + k % nx];
    return __syncthreads_or(any) != 0;
}
__device__ __forceinline__ bool lm_no_rare(const LmFrame& fr) { return (fr.deferred ? (uint32_t)fr.swap[7] : fr.counters[LM_CNT_RARE]) == 0u; }
// settle kernel of the between-frames flush (history export / import): the passes above ran iff the chain had turned
extern "C" __global__ void KN(lm_k_reuse_settle)(LmFrame fr)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && fr.swap[5] == 0 && (fr.swap[4] & 1) != (*fr.swap & 1)) fr.swap[5] = 1;
}

// What a resolved ReSTIR visibility ray does to its pixel: occluded => reservoir weight = 0 (WaveFrontShaders.cu:181-216); otherwise the reservoir is shaded
// into DIRECT with weight / 3 (ShadeInternal / ShadeReservoirs, ReSTIRKernels.cu:600-665).  One place: the two traversal kernels below and the
// known-answer hook lm_k_kat_resolve (which takes `occluded` from a mask instead of the tracer) run these lines.
__device__ __forceinline__ void lm_vis_resolve(const LmFrame& fr, int rc, float4* hot, uint32_t li, bool occluded, int pass)
{
    float* weight = (float*)(hot + 4u * li + 1u);           // quad 1 = (weight, count, normal.xy): lm_restir.h (included below)
    if (occluded) {
        // pass 2 with the frame's history passes pending (lazy reuse): the first spatial pass, which the reference runs BEFORE this one
        // (ReSTIR.cpp:181-212), will read this weight later — park it in the two spare words of quad 0 (read back in lm_restir_spatial_body, `parked`)
        if (pass == 2) { float* q0 = (float*)(hot + 4u * li); q0[2] = *weight; q0[3] = 1.f; }
        *weight = 0.f;
    } else {
        const lf3 add = v3(fr.resC[rc][li]) * (*weight / 3.f);
        float4 px = fr.direct[li];
        px.x += add.x; px.y += add.y; px.z += add.z;
        fr.direct[li] = px;
    }
}

// K6 + K23: resolve the visibility rays (tmin 0.1, WaveFrontShaders.cu:181-216: occluded => reservoir weight = 0) and shade
// the surviving reservoirs into DIRECT with weight / 3 (ReSTIRKernels.cu:600-665)
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_ANYHIT_WAVES)
KN(lm_k_restir_trace_shade)(LmScene sc, LmFrame fr, int rc, const uint32_t* __restrict__ countPtr, int refillBelow, int pass)
{
    rc = lm_res_idx(fr, rc);
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    const uint32_t n = *countPtr;
    float4* hot = fr.res[rc];
    const float4* __restrict__ qO = pass ? fr.vis2O : fr.visO;
    const float4* __restrict__ qD = pass ? fr.vis2D : fr.visD;
    lm_trace_queue<true>(sc, n, refillBelow, lm_make_stack(s_stack, sc), lm_stage_top(s_top, sc), fr.counters,
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) { const float4 o4 = qO[i]; o = v3(o4); d = v3(qD[i]); t0 = 0.1f; t1 = o4.w; },
        [&](uint32_t i, bool occluded, const LmHit&) { lm_vis_resolve(fr, rc, hot, f2u(qD[i].w), occluded, pass); });
}

// the same pass for queues of COHERENT visibility rays: the candidate pick / temporal pass append a 16 x 16 pixel tile's rays together, a
// wavefront's 64 rays start on neighbouring surface points and (with few lights) head for the same emitter, so the wavefront walks the
// tree as one (lm_trace_packets: shared stack, no per-lane ordering or stack traffic).  Occlusion is a yes / no answer: identical results.
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_TRACE_WAVES)
KN(lm_k_restir_trace_shade_packet)(LmScene sc, LmFrame fr, int rc, const uint32_t* __restrict__ countPtr, int pass)
{
    rc = lm_res_idx(fr, rc);
    __shared__ int s_wstack[LM_PACKET_STACK * (LM_BLOCK / 64)];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    const uint32_t n = *countPtr;
    float4* hot = fr.res[rc];
    const float4* __restrict__ qO = pass ? fr.vis2O : fr.visO;
    const float4* __restrict__ qD = pass ? fr.vis2D : fr.visD;
    lm_trace_packets<true>(sc, n, (lm_lds_int*)(s_wstack + LM_PACKET_STACK * (threadIdx.x >> 6)), lm_stage_top(s_top, sc),
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) { const float4 o4 = qO[i]; o = v3(o4); d = v3(qD[i]); t0 = 0.1f; t1 = o4.w; },
        [&](uint32_t i, bool occluded, const LmHit&) { lm_vis_resolve(fr, rc, hot, f2u(qD[i].w), occluded, pass); });
}

#include "lm_restir.h"
// Issue priority of the EXACT instantiations of the ReSTIR history kernels (bit 0 spatial, 1 temporal, 2 combine).  In the exact mode the ReSTIR stream is the frame's critical chain
// and its kernels are long VALU kernels (IEEE division / square-root sequences) that otherwise queue behind the wave chain's priority-3 kernels on every SIMD; the fast instantiations
// are memory-bound and do not respond (profiles/r06_prio_ab.txt).
#ifndef LM_EXACT_RESTIR_PRIO
#define LM_EXACT_RESTIR_PRIO 1
#endif

// K20 FillLightBags — ReSTIRKernels.cu:343-370
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_fill_bags)(LmScene sc, LmFrame fr, uint32_t seed, uint32_t total)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= total) return;
    uint32_t s = lm_wang_hash(seed + lm_wang_hash(i));
    const float rnd = lm_random_float(s);
    uint32_t li; float pdf;
    lm_cdf_get(sc, rnd, li, pdf);
    fr.bags[i] = make_uint2(li, f2u(pdf));
}

// K21 PickPrimarySamples — ReSTIRKernels.cu:402-522.  One 16x16 pixel tile (aligned to the GLOBAL 16x16 grid) shares a light
// bag (the reference keys the bag on the hardware SM id, which is not reproducible: DESIGN.md decision D2); the bag's 1000
// (index, pdf) pairs are staged in LDS once per tile.
// LDSL: the scene's light table fits LM_PICK_LDS_LIGHTS records and is staged in LDS as (p0, arm1, arm2, normal, radiance, area) — the 32
// candidates of a pixel then read their light with four ds_read_b128 instead of four global gathers (the texture path is what this
// kernel and the traversal kernels beside it compete for), and the arms are subtracted once per tile instead of once per candidate.
#ifndef LM_PICK_LDS_LIGHTS
#define LM_PICK_LDS_LIGHTS 384u    // table size = the largest light list that takes the LDS path, AND the kernel's residency cap: 8 KB bag + 24 KB table = five blocks per CU.
#endif                             // The pick saturates the vector ALUs; at six blocks (256 records) it holds more of every CU against the other streams' kernels than it can use:
                                   // 256 / 384 / 512 / 768 records = 6 / 5 / 4 / 3 blocks per CU -> 2584 / 2615 / 2469 / 2328 Mrays/s (six runs each, builds interleaved on one box,
                                   // profiles/r03_pick_residency_ab.txt)
#ifndef LM_PICK_STATIC_LDS
#define LM_PICK_STATIC_LDS 1       // 1: the table is a static 16-KB array whatever the light count; 0: sized by the launch (128 B for the benchmark scene's two lights).
#endif                             // The smaller footprint lets more blocks of this VALU-saturating kernel onto a CU and the frame loses 0.7 % (2551 -> 2533, six runs each,
                                   // builds interleaved on one box): the 16 KB double as the residency cap the other streams' kernels need.
// WIDE: a 1024-thread block takes FOUR tiles (one per 256 threads, each with its own bag) and shares one light table — see lm_k_pick_primary*_wide below.
template <class A, int ROLE, bool LDSL = false, bool WIDE = false>
__device__ __forceinline__ void lm_pick_primary_body(const LmScene& sc, const LmFrame& fr, int cur, int rc, uint32_t seed, uint32_t* visCount, uint2* s_bag, uint32_t* s_tmp,
                                                     float4* s_lights = nullptr, uint32_t vb = blockIdx.x, bool tileOk = true)
{
    const uint32_t tid = WIDE ? threadIdx.x & (LM_BLOCK - 1u) : threadIdx.x;      // thread within its tile
    if (ROLE == LM_RARE && fr.counters[LM_CNT_RARE] == 0u) return;      // grid-uniform: no surface of this frame needs the second launch
#ifdef LM_PICK_PRIO
    __builtin_amdgcn_s_setprio(LM_PICK_PRIO);          // the pick sits on the frame's critical chain (eager history passes) and is pure VALU: it must not be the kernel that waits (r06_pick_prio_ab.txt)
#endif
    if constexpr (LDSL) {
        for (uint32_t k = threadIdx.x; k < sc.numLights; k += (WIDE ? 4u * LM_BLOCK : LM_BLOCK)) {
            const LmTriLight l = lm_load_light(sc.lights, k);
            const lf3 arm1 = l.p1 - l.p0, arm2 = l.p2 - l.p0;
            s_lights[4u * k] = make_float4(l.p0.x, l.p0.y, l.p0.z, arm1.x);
            s_lights[4u * k + 1u] = make_float4(arm1.y, arm1.z, arm2.x, arm2.y);
            s_lights[4u * k + 2u] = make_float4(arm2.z, l.normal.x, l.normal.y, l.normal.z);
            s_lights[4u * k + 3u] = make_float4(l.radiance.x, l.radiance.y, l.radiance.z, l.area);
        }
    }
    rc = lm_res_idx(fr, rc);
    const uint32_t tilesX = (fr.W + 15u) / 16u;
    const uint32_t tx0 = fr.x0 / 16u, ty0 = fr.y0 / 16u;
    const uint32_t wtx = (fr.x0 + fr.ww + 15u) / 16u - tx0;
    const uint32_t tileX = tx0 + vb % wtx, tileY = ty0 + vb / wtx;
    if constexpr (ROLE == LM_RARE) {                                // this block's pixels (a tile of the GLOBAL grid) in window-local coordinates
        const int bx = (int)(tileX * 16u) - (int)fr.x0, by = (int)(tileY * 16u) - (int)fr.y0;
        if (!lm_rare_near(fr, cur, bx, by, bx + 15, by + 15)) return;
    }
    uint32_t bagSeed = lm_wang_hash(seed + (tileY * tilesX + tileX));
    const float rb = lm_random_float(bagSeed);
    const int bagIndex = (int)roundf((float)(50 - 1) * rb);
    if (tileOk) for (uint32_t k = tid; k < 1000u; k += LM_BLOCK) {
        uint2 e = fr.bags[(uint32_t)bagIndex * 1000u + k];
        if constexpr (A::contracted) e.y = f2u(A::rcp(u2f(e.y)));      // the loop multiplies by 1 / pdf: 1 000 reciprocals per tile instead of 8 192
        s_bag[k] = e;
    }
    __syncthreads();
    const uint32_t px = tileX * 16u + (tid & 15u), py = tileY * 16u + (tid >> 4);
    const bool inside = tileOk && !(px < fr.x0 || py < fr.y0 || px >= fr.x0 + fr.ww || py >= fr.y0 + fr.wh);
    uint32_t li = 0;
    bool shoot = false;
    lf3 vpos = v3(0.f), vdir = v3(0.f);
    float vlen = 0.f;
    if (inside) {
        li = (py - fr.y0) * fr.ww + (px - fr.x0);
        const uint32_t gi = py * fr.W + px;
        float4* hot = fr.res[rc];
        LmSurface pixel;
        lm_gbuf_load(fr.gbuf[cur], li, pixel);
        if (pixel.flags) { if (ROLE != LM_RARE) lm_hot_zero_weight(hot, li); }
        else if (lm_role_takes<ROLE>(pixel.mat)) {
            uint32_t s = lm_wang_hash(seed + lm_wang_hash(gi));
            LmReservoir fresh; lm_res_fresh(fresh);
            LmTarget target;                                       // the pixel's surface, prepared once for its 32 candidates
            lm_target_setup<A>(pixel, target);
            // one candidate from RNG state `st`: a light from the tile's bag, a point on it, its score at this surface; returns the resampling weight
            auto candidate = [&](uint32_t& st, LmSample& cand) -> float {
                const float r = lm_random_float(st);
                const int pick = lm_round_nonneg((float)(1000 - 1) * r);
                const uint2 entry = s_bag[pick];
                const float initialPdf = u2f(entry.y);               // contracted policy: its reciprocal (see the staging loop)
                const float u = lm_random_float(st);
                const float v = lm_random_float(st) * (1.f - u);
                lf3 p0, arm1, arm2;
                if constexpr (LDSL) {
                    const lm_lds_u4* lp = (const lm_lds_u4*)s_lights + 4u * entry.x;
                    const uint4 a = lm_lds_read4(lp), b = lm_lds_read4(lp + 1), c = lm_lds_read4(lp + 2), d = lm_lds_read4(lp + 3);
                    p0 = v3(u2f(a.x), u2f(a.y), u2f(a.z)); arm1 = v3(u2f(a.w), u2f(b.x), u2f(b.y)); arm2 = v3(u2f(b.z), u2f(b.w), u2f(c.x));
                    cand.p.normal = v3(u2f(c.y), u2f(c.z), u2f(c.w)); cand.p.radiance = v3(u2f(d.x), u2f(d.y), u2f(d.z)); cand.p.area = u2f(d.w);
                } else {
                    const LmTriLight light = lm_load_light(sc.lights, entry.x);
                    cand.p.radiance = light.radiance; cand.p.normal = light.normal; cand.p.area = light.area;
                    p0 = light.p0; arm1 = light.p1 - light.p0; arm2 = light.p2 - light.p0;
                }
                cand.p.position = p0 + (arm1 * u) + (arm2 * v);
                cand.contribution = v3(0.f);
                lm_score<A>(cand.p, target, cand.contribution, cand.pdf);
                return A::contracted ? cand.pdf * initialPdf : A::div(cand.pdf, initialPdf);
            };
            // The loop only remembers WHERE in the RNG stream the candidate the reservoir holds began (one register instead of the 14 of a
            // sample, which a take by any lane of the wavefront would copy); that candidate is generated once more behind the loop.
            uint32_t heldAt = 0u;
            bool holds = false;
            for (int smp = 0; smp < 32; smp++) {
                const uint32_t begin = s;
                LmSample cand;
                const float w = candidate(s, cand);
                if (lm_res_update_decide<A>(fresh, w, s)) { heldAt = begin; holds = true; }
            }
            if (holds) (void)candidate(heldAt, fresh.s);
            lm_res_update_weight<A>(fresh);
            lm_res_store(hot, fr.resC[rc], li, fresh);
            // K22 GenerateShadowRay fused (ReSTIRKernels.cu:546-582): the visibility ray of the fresh reservoir
            if (fresh.weight > 0.f) {
                vpos = pixel.position;
                vdir = fresh.s.p.position - vpos;
                vlen = length3(vdir);
                vdir = vdir / vlen;
                shoot = true;
            }
        }
    }
    const uint32_t slot = lm_append_slot_block(visCount, shoot, s_tmp);
    if (shoot) {
        fr.visO[slot] = v4(vpos, vlen - 0.05f);
        fr.visD[slot] = v4(vdir, u2f(li));
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_pick_primary)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    lm_pick_primary_body<LmExact, LM_ALL>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_FAST_WAVES)
KN(lm_k_pick_primary_fast)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
#ifdef LM_PICK_LDS_PAD
    __shared__ uint32_t s_pad[LM_PICK_LDS_PAD / 4];      // occupancy cap (A/B knob): fewer resident blocks of this VALU-saturating kernel
    if (seed == 0xffffffffu && visCount == nullptr) s_pad[threadIdx.x] = 1u;
#endif
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    lm_pick_primary_body<LmFast, LM_COMMON>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp);
}
// the same two kernels for scenes whose light table fits in LDS
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_pick_primary_lds)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
#if LM_PICK_STATIC_LDS
    __shared__ float4 s_lights[4 * LM_PICK_LDS_LIGHTS + 1];      // A/B switch: the 16-KB table of round 2
#else
    extern __shared__ float4 s_lights[];             // 64 B per emissive triangle, sized by the launch (l_pick_primary): the benchmark scene's two lights cost 128 B, not 16 KB
#endif
    lm_pick_primary_body<LmExact, LM_ALL, true>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp, s_lights);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_FAST_WAVES)
KN(lm_k_pick_primary_fast_lds)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
#if LM_PICK_STATIC_LDS
    __shared__ float4 s_lights[4 * LM_PICK_LDS_LIGHTS + 1];
#else
    extern __shared__ float4 s_lights[];
#endif
    lm_pick_primary_body<LmFast, LM_COMMON, true>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp, s_lights);
}
// ... and for light lists of LM_PICK_LDS_LIGHTS + 1 .. LM_PICK_LDS_LIGHTS_BIG records (round 5: the reference's default model, LowpolyRoom, has 414 triangle lights and fell
// off the 384-record table onto the global-gather path): a 32-KB table, four blocks per CU.  A/B on one box, interleaved (profiles/r05_pick_lds_big_ab.txt): LowpolyRoom
// 2 418 -> 2 480 Mrays/s fast (+2.5 %), 2 205 -> 2 274 exact (+3.1 %); scenes at or below 384 lights keep the smaller table (C2 loses 5 % at four blocks per CU, above).
#ifndef LM_PICK_LDS_LIGHTS_BIG
#define LM_PICK_LDS_LIGHTS_BIG 512u
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_pick_primary_lds_big)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    __shared__ float4 s_lights[4 * LM_PICK_LDS_LIGHTS_BIG + 1];
    lm_pick_primary_body<LmExact, LM_ALL, true>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp, s_lights);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, 4)      // 40 KB of LDS per block: four blocks = four waves per SIMD is what fits
KN(lm_k_pick_primary_fast_lds_big)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    __shared__ float4 s_lights[4 * LM_PICK_LDS_LIGHTS_BIG + 1];
    lm_pick_primary_body<LmFast, LM_COMMON, true>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp, s_lights);
}
// ... and for LM_PICK_LDS_LIGHTS_BIG + 1 .. LM_PICK_WIDE_LIGHTS records (round 6; BASELINE config C3: 513 records, one per quad of its 1 026 emissive triangles): a table
// of that size leaves room for one or two 256-thread blocks per CU, and the pick needs its four waves per SIMD (a 1 040-record table at two blocks per CU lost 4.6 %, LOG
// round 5).  So the BLOCK grows instead: 1 024 threads = four tiles, each quarter with its own 8-KB bag, all sixteen waves reading ONE table of 64 B x numLights (dynamic
// LDS: 32 KB + 64 B per light; C3 = 65 KB, two blocks per CU; from 1 000 lights on one block per CU = four waves per SIMD).  Same candidates, same arithmetic: the image is
// the one of the global-gather kernel.  A quarter whose tile index lies behind the last tile (tile count not a multiple of four) stages nothing and joins the barriers.
#ifndef LM_PICK_WIDE_LIGHTS
#define LM_PICK_WIDE_LIGHTS 1984u      // 32 000 + 68 + 64 x 1 984 + 16 = 159 060 B of the CU's 163 840
#endif
extern "C" __global__ void __launch_bounds__(4 * LM_BLOCK, 1)
KN(lm_k_pick_primary_wide)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount, uint32_t tiles)
{
    __shared__ uint2 s_bag[4000];
    __shared__ uint32_t s_tmp[17];
    extern __shared__ float4 s_lights[];
    const uint32_t vb = 4u * blockIdx.x + (threadIdx.x >> 8);
    lm_pick_primary_body<LmExact, LM_ALL, true, true>(sc, fr, cur, rc, seed, visCount, s_bag + 1000u * (threadIdx.x >> 8), s_tmp, s_lights, min(vb, tiles - 1u), vb < tiles);
}
extern "C" __global__ void __launch_bounds__(4 * LM_BLOCK, 1)
KN(lm_k_pick_primary_fast_wide)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount, uint32_t tiles)
{
    __shared__ uint2 s_bag[4000];
    __shared__ uint32_t s_tmp[17];
    extern __shared__ float4 s_lights[];
    const uint32_t vb = 4u * blockIdx.x + (threadIdx.x >> 8);
    lm_pick_primary_body<LmFast, LM_COMMON, true, true>(sc, fr, cur, rc, seed, visCount, s_bag + 1000u * (threadIdx.x >> 8), s_tmp, s_lights, min(vb, tiles - 1u), vb < tiles);
}
// LM_PICK_PERSIST = N > 0 (round 4): the same kernel as a PERSISTENT grid of N blocks per CU that loops over the tiles, with the light table sized by the launch.  The
// residency cap of the static table (five blocks per CU, above) is then the grid's size and no longer 24 KB of LDS per block that nobody reads: 8 KB + 64 B per light stay,
// and the kernels of the other streams (19.5 KB per block of the shading kernels, 17.7 KB of the traversal kernels) find room on the CU while the pick runs.
#ifndef LM_PICK_PERSIST
#define LM_PICK_PERSIST 0
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_FAST_WAVES)
KN(lm_k_pick_primary_fast_lds_persist)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount, uint32_t tiles)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    extern __shared__ float4 s_lights[];
    for (uint32_t vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
        lm_pick_primary_body<LmFast, LM_COMMON, true>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp, s_lights, vb);
        __syncthreads();                                        // the next tile's bag overwrites s_bag / s_tmp
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_pick_primary_rare)(LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint2 s_bag[1000];
    __shared__ uint32_t s_tmp[5];
    lm_pick_primary_body<LmExact, LM_RARE>(sc, fr, cur, rc, seed, visCount, s_bag, s_tmp);
}

// K24 temporal reuse.  `rf` = where this frame's fresh candidates live: the current buffer `rc` itself, or — when candidate
// generation of the NEXT frame runs ahead on its own stream — a separate buffer, so that it does not have to wait for this
// frame's spatial passes; the result lands in `rc` either way.
// K24 temporal reuse — ReSTIRKernels.cu:1015-1121
template <class A, int ROLE>
__device__ __forceinline__ void lm_restir_temporal_body(const LmFrame& fr, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount, uint32_t* s_tmp)
{
    if (ROLE == LM_RARE && fr.counters[LM_CNT_RARE] == 0u) return;
    if constexpr (!A::contracted && (LM_EXACT_RESTIR_PRIO & 2)) __builtin_amdgcn_s_setprio(3);
    // the previous frame's pending history passes were launched before this kernel and have run or returned: nothing is owed any more (lm_reuse_owed)
    if (ROLE != LM_RARE && blockIdx.x == 0 && threadIdx.x == 0) fr.swap[5] = 1;
    rc = lm_res_idx(fr, rc);
    rp = lm_res_idx(fr, rp);
    rf = lm_res_idx(fr, rf);
    if constexpr (ROLE == LM_RARE) {                                // the role follows the pixel's own current surface
        uint32_t ttx, tty; lm_tile_origin(fr, ttx, tty);
        if (!lm_rare_near(fr, cur, (int)(ttx << 4), (int)(tty << 4), (int)(ttx << 4) + 15, (int)(tty << 4) + 15)) return;
    }
    uint32_t li = 0, gi = 0;
    const bool valid = lm_tile_pixel(fr, li, gi);
    bool shoot = false;
    lf3 vpos = v3(0.f), vtarget = v3(1.f);
    if (valid) {
        const float4 cn = fr.probe[cur][li];
        if (cn.w >= 0.f) {                                         // unflagged current surface
            const int ly = (int)(li / fr.ww), lx = (int)(li - (uint32_t)ly * fr.ww);
            const uint32_t mv = fr.motion[li];
            const float vx = lm_f16_to_f32(mv & 0xffffu), vy = lm_f16_to_f32(mv >> 16);
            const int movedX = (int)roundf((float)fr.W * vx), movedY = (int)roundf((float)fr.H * vy);
            const int ty = ly + movedY, tx = lx + movedX;
            uint32_t tli = li;
            if (ty >= 0 && ty < (int)fr.wh && tx >= 0 && tx < (int)fr.ww) tli = (uint32_t)ty * fr.ww + (uint32_t)tx;
            const float4 pn = fr.probe[prev][tli];
            bool merged = false, mine = false;
            float weight = 0.f;
            if (pn.w >= 0.f) {
                const float d1 = pn.w, d2 = cn.w;
                const float depthDif = fabsf(d1 - d2) / ((d1 + d2) / 2.f);
                const float angle = dot3(v3(pn), v3(cn));
                if (depthDif < 0.10f && angle > 0.72222222223f) {
                    merged = true;
#ifndef LM_TEMPORAL_SHORTCUT
#define LM_TEMPORAL_SHORTCUT 1     // 0: A/B switch, the previous reservoir is always gathered and both samples re-evaluated
#endif
                    const bool prevLive = !LM_TEMPORAL_SHORTCUT || rp > 1 || fr.swap[2 + rp] != 0;       // block-uniform
                    LmSurface s;
                    if (prevLive) lm_gbuf_load(fr.gbuf[cur], li, s);
                    else {                                         // the shortcut below needs the position and the packed parameters only: the first half of the record's line
                        const float4 q0 = fr.gbuf[cur][8u * li + LM_GB_POSITION], q7 = fr.gbuf[cur][8u * li + LM_GB_PARAMS];
                        s.position = v3(q0); s.mat.p0 = f2u(q7.x); s.mat.p1 = f2u(q7.y); s.mat.p2 = f2u(q7.z);
                    }
#ifndef LM_TEMPORAL_INPLACE
#define LM_TEMPORAL_INPLACE 1      // 0: A/B switch, the shortcut loads and stores the whole reservoir
#endif
                    if (LM_TEMPORAL_INPLACE && !prevLive && rf == rc && lm_role_takes<ROLE>(s.mat)) {
                        // The same shortcut IN PLACE (candidates picked on this stream: the fresh reservoir already sits in the current buffer).  The two Updates
                        // below decide from numbers alone whether the fresh sample is kept, so the sample itself — position, normal, radiance, contribution —
                        // is neither loaded nor rewritten: quads 0 and 1 in, (weightSum, pdf) and the weight out; the light point only to aim the ray.
                        mine = true;
                        float4* h = fr.res[rc] + 4u * li;
                        const float4 h0 = h[0], h1 = h[1];
                        const uint32_t sd = lm_wang_hash(seed + gi);
                        LmReservoir out; lm_res_fresh(out);
                        (void)lm_res_update_decide<A>(out, (float)0ll * 0.f * 0.f, sd);
                        const long long cnt = lm_hot_count(h1);
                        if (lm_res_update_decide<A>(out, (float)cnt * lm_hot_weight(h1) * h0.y, sd)) {
                            out.s.pdf = h0.y;
                            out.count = 0ll + cnt;
                            lm_res_update_weight<A>(out);
                            h[0] = make_float4(out.weightSum, out.s.pdf, 0.f, 0.f);
                            *lm_hot_weight_at(fr.res[rc], li) = out.weight;
                            if (out.weight > 0.f) vtarget = v3(h[3]);
                        } else {                                   // (weight 0 after the first visibility pass, or a degenerate product): the zero sample survives
                            out.count = 0ll + cnt;
                            lm_res_update_weight<A>(out);
                            lm_res_store(fr.res[rc], fr.resC[rc], li, out);
                            vtarget = out.s.p.position;
                        }
                        weight = out.weight; vpos = s.position;
                    } else if (lm_role_takes<ROLE>(s.mat)) {       // fast mode: the other launch merges this pixel
                        mine = true;
                        LmReservoir rpv, rcv, out;
                        lm_res_load(fr.res[rf], fr.resC[rf], li, rcv);
                        if (prevLive) {
                            lm_res_load(fr.res[rp], fr.resC[rp], tli, rpv);
                            if (rpv.weight > 0.f) {                    // ShadeReservoirs on the PREVIOUS reservoir
                                const lf3 add = rpv.s.contribution * (rpv.weight / 3.f);
                                float4 px = fr.direct[li];
                                px.x += add.x; px.y += add.y; px.z += add.z;
                                fr.direct[li] = px;
                            }
                            const long long cap = rcv.count * 20;
                            if (cap < rpv.count) rpv.count = cap;
                            LmTarget target;
                            lm_target_setup<A>(s, target);
                            if (LM_TEMPORAL_SHORTCUT) lm_combine2_b_scored_here<A>(out, rpv, rcv, target, lm_wang_hash(seed + gi));     // the fresh sample was scored at this surface by the pick
                            else lm_combine2<A>(out, rpv, rcv, target, lm_wang_hash(seed + gi));
                        } else {
                            // The previous buffer has not been written since the reservoirs were reset (every frame of an even path depth: the
                            // reference's swap quirk): it holds the reset reservoir — count 0, weight 0, a zero sample.  CombineBiased of that and the
                            // fresh reservoir needs neither gather nor evaluation: Resample of the zero sample ends at cosOut <= 0 (pdf 0, weight
                            // (float)0 * 0 * 0), and Resample of the fresh sample at THIS surface returns what the candidate pick stored — the
                            // same function of the same surface record and light point.  The two Updates, the count and UpdateWeight run as in
                            // lm_combine2, so the reservoir written is the same, bit for bit (the whole parity suite runs through this branch).
                            const uint32_t sd = lm_wang_hash(seed + gi);
                            lm_res_fresh(out);
                            LmSample zero; lm_sample_zero(zero);
                            lm_res_update<A>(out, zero, (float)0ll * 0.f * 0.f, sd);
                            lm_res_update<A>(out, rcv.s, (float)rcv.count * rcv.weight * rcv.s.pdf, sd);
                            out.count = 0ll + rcv.count;
                            lm_res_update_weight<A>(out);
                        }
                        lm_res_store(fr.res[rc], fr.resC[rc], li, out);
                        weight = out.weight; vtarget = out.s.p.position; vpos = s.position;
                    }
                }
            }
            if (!merged && ROLE != LM_RARE) {                      // the fresh reservoir becomes the current one unchanged (no evaluation: first launch)
                mine = true;
                const float4* h = fr.res[rf] + 4u * li;
                const float4 h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
                if (rf != rc) {
                    float4* o = fr.res[rc] + 4u * li;
                    o[0] = h0; o[1] = h1; o[2] = h2; o[3] = h3;
                    fr.resC[rc][li] = fr.resC[rf][li];
                }
                weight = ((const float*)h)[4];                      // quad 1 = (weight, count, normal.xy): lm_restir.h (its own load: taking h1 apart
                                                                   // makes the compiler park the other three words in LDS)
                if (weight > 0.f) { vtarget = v3(h3); vpos = v3(fr.gbuf[cur][8u * li]); }
            }
            shoot = mine && weight > 0.f && lm_owned(fr, li, 0);   // second GenerateShadowRay pass (ReSTIR.cpp:211), fused; only owned pixels are combined
        } else if (rf != rc && ROLE != LM_RARE) {
            // flagged pixel: the candidate pick only zeroes the weight of the CURRENT reservoir and leaves the rest stale
            // (ReSTIRKernels.cu:441-447); when the pick wrote to its own buffer, do that here — a later frame may read this
            // entry as "previous" through a probe plane of a different age
            lm_hot_zero_weight(fr.res[rc], li);
        }
    }
    lf3 vdir = vtarget - vpos;
    const float vlen = length3(vdir);
    vdir = vdir / vlen;
    const uint32_t slot = lm_append_slot_block(visCount, shoot, s_tmp);
    if (shoot) {
        fr.vis2O[slot] = v4(vpos, vlen - 0.05f);
        fr.vis2D[slot] = v4(vdir, u2f(li));
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_restir_temporal)(LmFrame fr, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint32_t s_tmp[5];
    lm_restir_temporal_body<LmExact, LM_ALL>(fr, cur, prev, rc, rp, rf, seed, visCount, s_tmp);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_FAST_WAVES)
KN(lm_k_restir_temporal_fast)(LmFrame fr, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint32_t s_tmp[5];
    lm_restir_temporal_body<LmFast, LM_COMMON>(fr, cur, prev, rc, rp, rf, seed, visCount, s_tmp);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_restir_temporal_rare)(LmFrame fr, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount)
{
    __shared__ uint32_t s_tmp[5];
    lm_restir_temporal_body<LmExact, LM_RARE>(fr, cur, prev, rc, rp, rf, seed, visCount, s_tmp);
}

// K25 spatial reuse — ReSTIRKernels.cu:787-980 (biased branch).  The five candidate probes are issued together (one
// 16-byte gather each from the probe plane) and the accepted candidates' 64-byte reservoir records are fetched one
// iteration ahead of their re-evaluation, so the kernel is not a chain of dependent L2 round trips.
#ifndef LM_SPATIAL_WAVES
#define LM_SPATIAL_WAVES 3      // <= 168 VGPRs: three waves per SIMD instead of two (the kernel is gather-latency bound)
#endif
// LDS_PROBES (first pass only, 1024-thread blocks = 32 x 32 pixel tiles): the block stages the probes of its tile grown by the 30-pixel reach of
// the neighbour draw — 92 x 92 x 16 B = 132 KB of the CU's 160 KB LDS — with coalesced loads, and the five similarity tests of a pixel read LDS
// instead of gathering five 16-byte probes through the L1.  The probes are the same bits, so the verdicts and the image are unchanged.
#define LM_SPATIAL_WIN ((1u << LOG_TS) + 60u)
// FUSE (second pass, eager frames without a second-launch role): the pass ENDS with the pixel's CombineReservoirBuffers (K26, lm_restir_combine_body below) — the reservoir it
// has just written stays in registers as the combine's second operand instead of being read back by another full-screen launch.  The spatial output is still stored (a later
// frame's Reset() keeps "the sample that was there"), the arithmetic and its order are the combine kernel's: same bits.
template <class A, int ROLE>
__device__ __forceinline__ void lm_fused_combine(const LmFrame& fr, int cur, uint32_t li, uint32_t gi, const LmReservoir& b)
{
    LmTarget target;
    {
        LmSurface s;
        lm_gbuf_load(fr.gbuf[cur], li, s);
        if (!lm_role_takes<ROLE>(s.mat)) return;
        lm_target_setup<A>(s, target);
    }
    const int rc = lm_res_idx(fr, fr.fuseRc);
    LmReservoir a, out;
    lm_res_load(fr.res[rc], fr.resC[rc], li, a);
    lm_combine2<A>(out, a, b, target, lm_wang_hash(fr.fuseSeed + gi));
    lm_res_store(fr.res[rc], fr.resC[rc], li, out);
}
template <class A, int ROLE, uint32_t LOG_TS = 4, bool LDS_PROBES = false, bool FUSE = false>
__device__ __forceinline__ void lm_restir_spatial_body(const LmFrame& fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass, float4* s_probe = nullptr, uint32_t vb = blockIdx.x)
{
    if (fr.deferred && !lm_reuse_owed(fr)) return;
    if (ROLE == LM_RARE && lm_no_rare(fr)) return;
    rin = lm_res_idx(fr, rin);
    rout = lm_res_idx(fr, rout);
    if constexpr (ROLE == LM_RARE) {                                // the role follows the FIRST accepted neighbour's surface: anywhere within the 30-pixel reach of the draw
        uint32_t ttx, tty; lm_tile_origin<LOG_TS>(fr, ttx, tty, vb);
        if (!lm_rare_near(fr, cur, (int)(ttx << LOG_TS) - 30, (int)(tty << LOG_TS) - 30, (int)(ttx << LOG_TS) + (1 << LOG_TS) + 29, (int)(tty << LOG_TS) + (1 << LOG_TS) + 29)) return;
    }
#ifdef LM_SPATIAL_PRIO
    __builtin_amdgcn_s_setprio(LM_SPATIAL_PRIO);
#else
    if constexpr (!A::contracted && (LM_EXACT_RESTIR_PRIO & 1)) __builtin_amdgcn_s_setprio(3);      // exact instantiation: LM_EXACT_RESTIR_PRIO (top of the ReSTIR section)
#endif
    int wx0 = 0, wy0 = 0;                                           // window origin of the staged probes (window-local pixels, may be negative)
    if constexpr (LDS_PROBES) {
        uint32_t tx, ty;
        lm_tile_origin<LOG_TS>(fr, tx, ty, vb);
        wx0 = (int)(tx << LOG_TS) - 30; wy0 = (int)(ty << LOG_TS) - 30;
        const float4* probe = fr.probe[cur];
        for (uint32_t k = threadIdx.x; k < LM_SPATIAL_WIN * LM_SPATIAL_WIN; k += blockDim.x) {
            const int py = wy0 + (int)(k / LM_SPATIAL_WIN), px = wx0 + (int)(k % LM_SPATIAL_WIN);
            if (px >= 0 && px < (int)fr.ww && py >= 0 && py < (int)fr.wh) s_probe[k] = probe[(uint32_t)py * fr.ww + (uint32_t)px];
        }
        __syncthreads();
    }
    uint32_t li = 0, gi = 0;
    if (!lm_tile_pixel<LOG_TS>(fr, li, gi, vb)) return;
    if (!lm_owned(fr, li, margin)) return;                       // pass 1 feeds pass 2 within 30 pixels of the owned tile, pass 2 only the tile
    const float4* hotIn = fr.res[rin];
    // The second pass draws the candidates of the first (same seed) against the same probe plane: the first pass leaves its verdicts
    // (5 bits, or "flagged pixel") in a 4-byte plane and the second reads that instead of gathering five probes again.
    uint32_t mask = 0u;
    if (pass) { mask = fr.reuseMask[li]; if (mask == LM_REUSE_FLAGGED) return; }
    const int y = (int)(li / fr.ww), x = (int)(li - (uint32_t)y * fr.ww);
    uint32_t s = lm_wang_hash(seed + gi);
    uint32_t cand[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int ny = (int)roundf((lm_random_float(s) * 2.f - 1.f) * 30.f) + y;
        const int nx = (int)roundf((lm_random_float(s) * 2.f - 1.f) * 30.f) + x;
        const bool in = !(nx < 0 || nx >= (int)fr.ww || ny < 0 || ny >= (int)fr.wh);
        cand[k] = in ? (uint32_t)ny * fr.ww + (uint32_t)nx : 0xffffffffu;
    }
    if (!pass) {
        const float4* probe = fr.probe[cur];
        float4 cn;
        float4 pr[5];
        if constexpr (LDS_PROBES) {
            auto at = [&](uint32_t pix) { const int py = (int)(pix / fr.ww), px = (int)(pix - (uint32_t)py * fr.ww); return s_probe[(uint32_t)(py - wy0) * LM_SPATIAL_WIN + (uint32_t)(px - wx0)]; };
            cn = at(li);
            if (cn.w < 0.f) { if (ROLE != LM_RARE) fr.reuseMask[li] = LM_REUSE_FLAGGED; return; }
#pragma unroll
            for (int k = 0; k < 5; k++) pr[k] = at(cand[k] != 0xffffffffu ? cand[k] : li);
        } else {
            cn = probe[li];
            if (cn.w < 0.f) { if (ROLE != LM_RARE) fr.reuseMask[li] = LM_REUSE_FLAGGED; return; }
#pragma unroll
            for (int k = 0; k < 5; k++) pr[k] = probe[cand[k] != 0xffffffffu ? cand[k] : li];
        }
        const float ct = cn.w;
        // accepted candidates as a bit mask; they are visited in candidate order (no dynamically indexed array: registers only)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (cand[k] == 0xffffffffu || pr[k].w < 0.f) continue;
            const float d1 = pr[k].w;
            const float depthDif = fabsf(d1 - ct) / ((d1 + ct) / 2.f);
            const float angle = dot3(v3(pr[k]), v3(cn));
            if (depthDif < 0.10f && angle > 0.72222222223f) mask |= 1u << k;
        }
        if (ROLE != LM_RARE) fr.reuseMask[li] = mask;
    }
    auto candAt = [&](uint32_t k) { return k == 0u ? cand[0] : k == 1u ? cand[1] : k == 2u ? cand[2] : k == 3u ? cand[3] : cand[4]; };
    float4* hotOut = fr.res[rout];
    LmReservoir fb; lm_res_fresh(fb);                               // FUSE: what this pass leaves at the pixel, the combine's second operand
    bool fused = false;
    if (__popc(mask) > 1) {
        const uint32_t nb0 = candAt((uint32_t)__ffs((int)mask) - 1u);
        mask &= mask - 1u;
        const float4* h = hotIn + 4u * nb0;
        float4 p1 = h[1], p2 = h[2], p3 = h[3];                    // the three quads a neighbour's reservoir is gathered by (lm_restir.h)
        // deferred first pass: the frame's second visibility pass has run meanwhile and zeroed occluded weights — after parking them (lm_k_restir_trace_shade)
        const bool parked = fr.deferred != 0 && pass == 0;
        if (parked) { const float2 z = *(const float2*)((const float*)h + 2); if (z.y == 1.f) p1.x = z.x; }
        LmTarget target;
        {
            LmSurface s0;
            lm_gbuf_load(fr.gbuf[cur], nb0, s0);                // every candidate is re-evaluated at the FIRST neighbour's surface (reference :883)
            if (!lm_role_takes<ROLE>(s0.mat)) return;           // fast mode: the other launch handles this pixel
            lm_target_setup<A>(s0, target);
        }
        LmReservoir out; lm_res_fresh(out);
        long long sum = 0;
        // (no prefetch of the next record: holding it costs 16 registers = one wave per SIMD, and five waves hide the gather as well: 500 us alone
        // either way, frame equal within noise — profiles/r02_spatial_prefetch_ab.txt)
        for (;;) {
            LmSample rs;
            rs.p = lm_point_unpack(p1, p2, p3);
            rs.contribution = v3(0.f);                             // the neighbour's own contribution is not carried over (reference: a fresh LightSample)
            const long long cnt = lm_hot_count(p1);
            lm_score<A>(rs.p, target, rs.contribution, rs.pdf);
            lm_res_update<A>(out, rs, (float)cnt * lm_hot_weight(p1) * rs.pdf, seed);   // global seed: same draw for all pixels (reference quirk)
            sum += cnt;
            if (mask == 0u) break;
            const float4* hn = hotIn + 4u * candAt((uint32_t)__ffs((int)mask) - 1u);
            mask &= mask - 1u;
            p1 = hn[1]; p2 = hn[2]; p3 = hn[3];
            if (parked) { const float2 z = *(const float2*)((const float*)hn + 2); if (z.y == 1.f) p1.x = z.x; }
        }
        out.count = sum;
        lm_res_update_weight<A>(out);
        lm_res_store(hotOut, fr.resC[rout], li, out);
        if constexpr (FUSE) { fb = out; fb.count = (long long)(uint32_t)out.count; fused = true; }      // (the count as the record holds it: 32 bits)
    } else if (ROLE != LM_RARE) {
        const float4 q0 = hotOut[4u * li], q1 = hotOut[4u * li + 1u];      // Reset(): weightSum, sampleCount, weight; the sample stays
        hotOut[4u * li] = make_float4(0.f, q0.y, 0.f, 0.f);
        hotOut[4u * li + 1u] = make_float4(0.f, u2f(0u), q1.z, q1.w);
        if constexpr (FUSE) {                                              // the combine's second operand = the record as it now stands: reset numbers, the old sample
            lm_res_unpack(make_float4(0.f, q0.y, 0.f, 0.f), make_float4(0.f, u2f(0u), q1.z, q1.w), hotOut[4u * li + 2u], hotOut[4u * li + 3u], fb);
            fb.s.contribution = v3(fr.resC[rout][li]);
            fused = true;
        }
    }
    if constexpr (FUSE) { if (fused) lm_fused_combine<A, ROLE>(fr, cur, li, gi, fb); }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SPATIAL_WAVES)
KN(lm_k_restir_spatial)(LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass) { lm_restir_spatial_body<LmExact, LM_ALL>(fr, cur, rin, rout, seed, margin, pass); }
#ifndef LM_SPATIAL_FAST_WAVES
#define LM_SPATIAL_FAST_WAVES 5
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SPATIAL_FAST_WAVES)
KN(lm_k_restir_spatial_fast)(LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass) { lm_restir_spatial_body<LmFast, LM_COMMON>(fr, cur, rin, rout, seed, margin, pass); }
// first pass with the probe window in LDS: one 1024-thread block per CU (132 KB), four waves per SIMD
extern "C" __global__ void __launch_bounds__(1024, 1)
KN(lm_k_restir_spatial_fast_lds)(LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin)
{
    __shared__ float4 s_probe[92u * 92u];
    lm_restir_spatial_body<LmFast, LM_COMMON, 5, true>(fr, cur, rin, rout, seed, margin, 0, s_probe);
}
// the same with the ordinary 16 x 16 tile (round 4): 76 x 76 probes = 92 KB, one 256-thread block per CU; spatial_lds = 2
extern "C" __global__ void __launch_bounds__(LM_BLOCK, 1)
KN(lm_k_restir_spatial_fast_lds16)(LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin)
{
    __shared__ float4 s_probe[76u * 76u];
    lm_restir_spatial_body<LmFast, LM_COMMON, 4, true>(fr, cur, rin, rout, seed, margin, 0, s_probe);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SPATIAL_WAVES)
KN(lm_k_restir_spatial_rare)(LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass) { lm_restir_spatial_body<LmExact, LM_RARE>(fr, cur, rin, rout, seed, margin, pass); }
// second pass + combine in one launch (tuning key fuse_combine; frame.cpp): the exact mode, and the fast mode of scenes without a second-launch material
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SPATIAL_WAVES)
KN(lm_k_restir_spatial_fused)(LmFrame fr, int cur, int rin, int rout, uint32_t seed) { lm_restir_spatial_body<LmExact, LM_ALL, 4, false, true>(fr, cur, rin, rout, seed, 0, 1); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_SPATIAL_FAST_WAVES)
KN(lm_k_restir_spatial_fast_fused)(LmFrame fr, int cur, int rin, int rout, uint32_t seed) { lm_restir_spatial_body<LmFast, LM_COMMON, 4, false, true>(fr, cur, rin, rout, seed, 0, 1); }

// K26 CombineReservoirBuffers — ReSTIRKernels.cu:1407-1436
template <class A, int ROLE>
__device__ __forceinline__ void lm_restir_combine_body(const LmFrame& fr, int cur, int rc, int rs, uint32_t seed, uint32_t vb = blockIdx.x)
{
    if (fr.deferred && !lm_reuse_owed(fr)) return;
    if (fr.deferred && ROLE != LM_RARE && vb == 0u && threadIdx.x == 0) ++fr.swap[8];             // statistic: deferred executions
    if (ROLE == LM_RARE && lm_no_rare(fr)) return;
    if constexpr (!A::contracted && (LM_EXACT_RESTIR_PRIO & 4)) __builtin_amdgcn_s_setprio(3);
    rc = lm_res_idx(fr, rc);
    rs = lm_res_idx(fr, rs);
    if constexpr (ROLE == LM_RARE) {                                // the role follows the pixel's own current surface
        uint32_t ttx, tty; lm_tile_origin(fr, ttx, tty, vb);
        if (!lm_rare_near(fr, cur, (int)(ttx << 4), (int)(tty << 4), (int)(ttx << 4) + 15, (int)(tty << 4) + 15)) return;
    }
    uint32_t li = 0, gi = 0;
    if (!lm_tile_pixel(fr, li, gi, vb)) return;
    if (!lm_owned(fr, li, 0)) return;
    if (fr.probe[cur][li].w < 0.f) return;
    LmTarget target;
    {
        LmSurface s;
        lm_gbuf_load(fr.gbuf[cur], li, s);
        if (!lm_role_takes<ROLE>(s.mat)) return;
        lm_target_setup<A>(s, target);
    }
    LmReservoir a, b, out;
    lm_res_load(fr.res[rc], fr.resC[rc], li, a);
    lm_res_load(fr.res[rs], fr.resC[rs], li, b);
#ifndef LM_COMBINE_SHORTCUT
#define LM_COMBINE_SHORTCUT 0      // 1: the current reservoir's sample, scored at this surface by the temporal pass, is not re-evaluated (lm_combine2_a_scored_here).
#endif                             //    Identical image, 6 M fewer wave instructions per frame — and the frame 0.5 % SLOWER (2454.7 -> 2443.3, same box, spread 0.1 %:
                                   //    profiles/r03_restir_shortcuts_ab.txt); the pass is an HBM stream.  Off.
    if (LM_COMBINE_SHORTCUT) lm_combine2_a_scored_here<A>(out, a, b, target, lm_wang_hash(seed + gi));
    else lm_combine2<A>(out, a, b, target, lm_wang_hash(seed + gi));
    lm_res_store(fr.res[rc], fr.resC[rc], li, out);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_restir_combine)(LmFrame fr, int cur, int rc, int rs, uint32_t seed) { lm_restir_combine_body<LmExact, LM_ALL>(fr, cur, rc, rs, seed); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_FAST_WAVES)
KN(lm_k_restir_combine_fast)(LmFrame fr, int cur, int rc, int rs, uint32_t seed) { lm_restir_combine_body<LmFast, LM_COMMON>(fr, cur, rc, rs, seed); }
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_RESTIR_WAVES)
KN(lm_k_restir_combine_rare)(LmFrame fr, int cur, int rc, int rs, uint32_t seed) { lm_restir_combine_body<LmExact, LM_RARE>(fr, cur, rc, rs, seed); }
// The same passes launched for the PREVIOUS frame (lazy reuse, lm_reuse_owed): a grid of at most 2048 blocks that loops over the tiles, so that the usual
// case — nothing owed, every block returns at once — costs the stream a few microseconds instead of 14 400 block dispatches among the resident kernels.
#define LM_DEFERRED(NAME, BOUNDS, BODY, PARAMS, ...) \
    extern "C" __global__ void __launch_bounds__(LM_BLOCK, BOUNDS) KN(NAME) PARAMS \
    { if (!lm_reuse_owed(fr)) return; for (uint32_t vb = blockIdx.x; vb < (uint32_t)tiles; vb += gridDim.x) BODY(__VA_ARGS__, vb); }
#define LM_SP_PARAMS (LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass, int tiles)
#define LM_CB_PARAMS (LmFrame fr, int cur, int rc, int rs, uint32_t seed, int tiles)
LM_DEFERRED(lm_k_restir_spatial_deferred, LM_SPATIAL_WAVES, (lm_restir_spatial_body<LmExact, LM_ALL>), LM_SP_PARAMS, fr, cur, rin, rout, seed, margin, pass, nullptr)
LM_DEFERRED(lm_k_restir_spatial_fast_deferred, LM_SPATIAL_FAST_WAVES, (lm_restir_spatial_body<LmFast, LM_COMMON>), LM_SP_PARAMS, fr, cur, rin, rout, seed, margin, pass, nullptr)
LM_DEFERRED(lm_k_restir_spatial_rare_deferred, LM_SPATIAL_WAVES, (lm_restir_spatial_body<LmExact, LM_RARE>), LM_SP_PARAMS, fr, cur, rin, rout, seed, margin, pass, nullptr)
LM_DEFERRED(lm_k_restir_combine_deferred, LM_RESTIR_WAVES, (lm_restir_combine_body<LmExact, LM_ALL>), LM_CB_PARAMS, fr, cur, rc, rs, seed)
LM_DEFERRED(lm_k_restir_combine_fast_deferred, LM_FAST_WAVES, (lm_restir_combine_body<LmFast, LM_COMMON>), LM_CB_PARAMS, fr, cur, rc, rs, seed)
LM_DEFERRED(lm_k_restir_combine_rare_deferred, LM_RESTIR_WAVES, (lm_restir_combine_body<LmExact, LM_RARE>), LM_CB_PARAMS, fr, cur, rc, rs, seed)

// Lazy reuse, the other half (see lm_reuse_owed).  The swap chain has NOT turned: the pending history passes of the previous frame are dropped, because
// this frame's candidate pick / temporal pass rewrites every entry of that buffer — except at pixels that were reuse surfaces then and are flagged now,
// where only the weight is zeroed.  A weight-0 entry is observable through ONE field: a later temporal pass adds its sampleCount to the merged count and
// caps it (ReSTIRKernels.cu:1062-1121; its sample is re-scored but enters every Update with weight count * 0 * pdf = 0, and its contribution is only
// shaded when the weight is positive).  The count CombineReservoirBuffers would have left there is temporal count + count of the second spatial
// pass = the sum, over that pass's accepted candidates, of the first pass's counts — sums of temporal counts over accepted candidates, zero where fewer
// than two are accepted (Reset).  Acceptance needs the probe plane only, so the cone (5 + 25 probes, 25 counts) costs no evaluation; it runs for the few
// silhouette pixels that sub-pixel jitter or motion turns into emitter / miss pixels.  `fo` = the previous frame's parameters, `now` = this frame's set.
__device__ __forceinline__ uint32_t lm_reuse_verdicts(const LmFrame& fr, const float4* __restrict__ probe, uint32_t li, uint32_t seed, uint32_t (&cand)[5])
{
    const int y = (int)(li / fr.ww), x = (int)(li - (uint32_t)y * fr.ww);
    const uint32_t gi = (fr.y0 + (uint32_t)y) * fr.W + (fr.x0 + (uint32_t)x);
    uint32_t s = lm_wang_hash(seed + gi);
    uint32_t mask = 0u;
    const float4 cn = probe[li];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int ny = (int)roundf((lm_random_float(s) * 2.f - 1.f) * 30.f) + y;
        const int nx = (int)roundf((lm_random_float(s) * 2.f - 1.f) * 30.f) + x;
        cand[k] = 0xffffffffu;
        if (nx < 0 || nx >= (int)fr.ww || ny < 0 || ny >= (int)fr.wh) continue;
        cand[k] = (uint32_t)ny * fr.ww + (uint32_t)nx;
        const float4 pr = probe[cand[k]];
        if (pr.w < 0.f) continue;
        const float depthDif = fabsf(pr.w - cn.w) / ((pr.w + cn.w) / 2.f);
        if (depthDif < 0.10f && dot3(v3(pr), v3(cn)) > 0.72222222223f) mask |= 1u << k;
    }
    return mask;
}
// two launches over the extraction's list (LmFrame::hazardList of THIS frame): phase 0 gathers the sums (the cone reads the counts of NEIGHBOURING entries,
// which may be due for completion themselves) into the reuse-mask plane, which is free while the passes that use it are dropped; phase 1 adds them
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_reuse_counts)(LmFrame fo, int was, const uint32_t* __restrict__ list, const uint32_t* __restrict__ listCount, uint32_t seed, int phase)
{
    if (fo.swap[5] != 0 || lm_reuse_turned(fo)) return;
    const float4* __restrict__ probe = fo.probe[was];
    float4* hot = fo.res[lm_res_idx(fo, LM_RES_OWED)];
    const uint32_t n = *listCount, stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t li = list[i];
        if (!lm_owned(fo, li, 0)) continue;                       // the combine did not write it
        if (phase) {
            float* cnt = (float*)(hot + 4u * li + 1u) + 1;
            *cnt = u2f(f2u(*cnt) + fo.reuseMask[li]);
            continue;
        }
        uint32_t c2[5];
        const uint32_t m2 = lm_reuse_verdicts(fo, probe, li, seed, c2);
        uint32_t sum2 = 0u;
        if (__popc(m2) > 1) {
            for (int j = 0; j < 5; j++) {
                if (!((m2 >> j) & 1u)) continue;
                uint32_t c1[5];
                const uint32_t m1 = lm_reuse_verdicts(fo, probe, c2[j], seed, c1);
                if (__popc(m1) > 1) for (int k = 0; k < 5; k++) if ((m1 >> k) & 1u) sum2 += f2u(hot[4u * c1[k] + 1u].y);
            }
        }
        fo.reuseMask[li] = sum2;
    }
    if (phase && blockIdx.x == 0 && threadIdx.x == 0) fo.swap[9] += (int)n;       // statistic: entries completed this way
}

#if LM_INSTRUMENT
extern "C" __global__ void lm_k_read_pushes(unsigned long long* out) { out[0] = g_lmPushes[0]; out[1] = g_lmPushes[1]; g_lmPushes[0] = 0; g_lmPushes[1] = 0; }
extern "C" void lm_read_pushes(hipStream_t s, unsigned long long* out) { hipLaunchKernelGGL(lm_k_read_pushes, dim3(1), dim3(1), 0, s, out); }
#endif
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_clear_f4)(float4* __restrict__ p, uint32_t n)
{
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// make_color (vendor/Include/Cuda/cuda/helpers.h:35-66): clamp, sRGB transfer function, quantise as x * 256 capped at 255
__device__ __forceinline__ uint32_t lm_srgb8(float x)
{
    const float in = clampf(x, 0.f, 1.f);
    const float powed = lm_powf(in, 1.0f / 2.4f);
    float sv = in < 0.0031308f ? 12.92f * in : 1.055f * powed - 0.055f;
    sv = clampf(sv, 0.f, 1.f);
    const uint32_t v = (uint32_t)(sv * 256.f);
    return v < 255u ? v : 255u;
}
// K13 + K14: channel merge with optional running-mean blend (GPUMergeOutputChannels.cu:5-88, fp32) and sRGB8 output
// (GPUShadingKernels.cu:28-56, vendor/Include/Cuda/cuda/helpers.h:35-66)
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_merge_output)(LmFrame fr, int blend, uint32_t blendCount, int depthMax)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {     // ReSTIR::SwapBuffers once per executed wave (WaveFrontRenderer.cpp:697,827)
        int executed = 0;
        for (int d = 0; d < depthMax; d++) executed += fr.counters[LM_CNT_RAYS(d)] > 0u;
        fr.swap[2 + (*fr.swap & 1)] = 1;             // this frame's passes wrote the current swap-chain buffer
        fr.swap[4] = *fr.swap & 1;                   // ... which is the one its history passes complete; blend bit 1: they are pending (lm_reuse_owed)
        fr.swap[5] = (blend & 2) ? 0 : 1;
        fr.swap[7] = (int)fr.counters[LM_CNT_RARE];
        *fr.swap = (*fr.swap + executed) & 1;
        fr.swap[1] = executed;
    }
    // running sums of the counter block (merges of consecutive frames are ordered on one stream, and every producer of this frame's
    // counters has been joined before the merge)
    if (blockIdx.x == 0 && threadIdx.x <= LM_CNT_WORDS) fr.totals[threadIdx.x] += threadIdx.x < LM_CNT_WORDS ? (unsigned long long)fr.counters[threadIdx.x] : 1ull;
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t li = blockIdx.x * LM_BLOCK + threadIdx.x; li < fr.n; li += stride) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        m = m + fr.direct[li];
        m = m + fr.indirect[li];
        float4 c;
        if (blend & 1) {
            const float4 old = fr.combined[li];
            const float k = (float)blendCount, k1 = (float)(blendCount + 1u);
            const float4 s = old * k + m;
            c = make_float4(s.x / k1, s.y / k1, s.z / k1, s.w / k1);
        } else c = m;
        fr.combined[li] = c;
        fr.output[li] = make_uchar4((unsigned char)lm_srgb8(c.x), (unsigned char)lm_srgb8(c.y), (unsigned char)lm_srgb8(c.z), 255);
    }
}

// K1 + K2-K4 for the primary wave in one launch: the packet kernel generates its rays itself (64 consecutive queue slots = one 8 x 8 pixel tile) and leaves
// the direction plane for the extraction (tuning key fuse_primary, off).  Alone lm_k_primary is a 27-us stream of 59 MB; at the head of a frame's wave
// chain, in a machine full of resident long-running blocks, it shows 200 us in the timeline (profiles/r03_c2_timeline_lazy_fast.txt) — but fusing it
// away changes nothing (- 0.3 %, profiles/r03_fuse_primary_ab.txt): the time was spent waiting for block slots, and the traversal waits for the same slots.
extern "C" __global__ void __launch_bounds__(LM_BLOCK, LM_TRACE_WAVES)
KN(lm_k_trace_primary_packet)(LmScene sc, LmFrame fr, LmCamera cam, uint32_t frameCount, uint4* __restrict__ hits, float tmin, float tmax)
{
    __shared__ int s_wstack[LM_PACKET_STACK * (LM_BLOCK / 64)];
    __shared__ uint4 s_top[LM_WIDTH * LM_TOP_NODES + 1];
    if (blockIdx.x == 0) {              // the frame's first kernel zeroes the frame's counter block (as lm_k_primary does); no other block of this kernel touches it
#if LM_PRIMARY_CLEARS
        for (uint32_t w = threadIdx.x; w < LM_CNT_WORDS; w += LM_BLOCK) fr.counters[w] = 0u;
        __syncthreads();
#endif
        if (threadIdx.x == 0) fr.counters[LM_CNT_RAYS(0)] = fr.n;
    }
    const lf3 eye = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    float4* __restrict__ plane = fr.rayD[0];
    lm_trace_packets<false>(sc, fr.n, (lm_lds_int*)(s_wstack + LM_PACKET_STACK * (threadIdx.x >> 6)), lm_stage_top(s_top, sc),
        [&](uint32_t i, lf3& o, lf3& d, float& t0, float& t1) {
            uint32_t li;
            o = eye; d = lm_primary_dir(fr, cam, frameCount, i, li); t0 = tmin; t1 = tmax;
            plane[i] = make_float4(d.x, d.y, d.z, u2f(li));
        },
        [&](uint32_t i, bool found, const LmHit& h) {
            uint4 out = make_uint4(0u, 0u, 0u, f2u(-1.f));
            if (found) {
                const uint2 id = sc.triId[h.slot];
                out.x = id.x; out.y = id.y;
                out.z = lm_f32_to_f16(h.u) | (lm_f32_to_f16(h.v) << 16);
                out.w = f2u(h.t);
            }
            hits[i] = out;
        });
}

// ---------------------------------------------------------------------------------------------------------------------
// test / seam kernels (the ray-query seam of OptixWrapper::TraceRays and batch BSDF evaluation for known-answer tests)
// ---------------------------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_query_any)(LmScene sc, const float4* __restrict__ rayO /* w = tmax */, const float4* __restrict__ rayD, uint32_t n, float tmin,
               uint32_t* __restrict__ occluded, uint32_t* counters)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    const LmStack stack = lm_make_stack(s_stack, sc);
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const float4 o4 = rayO[i], d4 = rayD[i];
        LmHit h;
        occluded[i] = lm_traverse<true>(sc, v3(o4), v3(d4), tmin, o4.w, stack, h, counters) ? 1u : 0u;
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_query_closest_raw)(LmScene sc, const float4* __restrict__ rayO, const float4* __restrict__ rayD, uint32_t n, float tmin, float tmax,
                       uint4* __restrict__ idOut, float4* __restrict__ uvtOut, uint32_t* counters)
{
    __shared__ int s_stack[LM_STACK_LDS * LM_BLOCK];
    const LmStack stack = lm_make_stack(s_stack, sc);
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        LmHit h; h.t = -1.f; h.u = 0.f; h.v = 0.f; h.slot = 0;
        const bool found = lm_traverse<false>(sc, v3(rayO[i]), v3(rayD[i]), tmin, tmax, stack, h, counters);
        uint2 id = make_uint2(0u, 0u);
        if (found) id = sc.triId[h.slot];
        idOut[i] = make_uint4(id.x, id.y, found ? 1u : 0u, 0u);
        uvtOut[i] = found ? make_float4(h.u, h.v, h.t, 0.f) : make_float4(0.f, 0.f, -1.f, 0.f);
    }
}
#ifndef LUMEN_MI_TEST_HOOKS
#define LUMEN_MI_TEST_HOOKS 1      // known-answer hooks (lm_hooks.h): on for the suite; `make HOOKS=0` builds the library without its test surface
#endif
#if LUMEN_MI_TEST_HOOKS
#define LM_HOOKS_PART 1
#include "lm_hooks.h"
#endif

// ---------------------------------------------------------------------------------------------------------------------
// launch table: this file is compiled twice (LM_INSTRUMENT = 0 / 1); the renderer picks a table at run time
// ---------------------------------------------------------------------------------------------------------------------
#include "lm_launch.h"
// Denoiser / upscaler inputs from the depth-0 G-buffer — reference GPUExtractNRD_DLSSdata.cu:6-89 (depth normalised to the
// camera's render-distance range, fp32; half4 normal + roughness), GPUExtractDepthData.cu:6-72.  Pixels without a hit
// (t < 0) get depth 0 and keep their previous normal-roughness value, as in the reference.
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_export_aux)(LmFrame fr, int cur, float minD, float maxD, float* __restrict__ depth, uint2* __restrict__ normalRoughness)
{
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t li = blockIdx.x * LM_BLOCK + threadIdx.x; li < fr.n; li += stride) {
        const float4* rec = fr.gbuf[cur] + 8u * li;
        const float4 a = rec[0], b = rec[1];
        const float t = a.w;
        if (t < 0.f) { if (depth) depth[li] = 0.f; continue; }
        if (depth) depth[li] = (t - fminf(minD, t)) / (fmaxf(maxD, t) - fminf(minD, t));
        if (normalRoughness) {
            LmMaterial m; m.p0 = f2u(rec[LM_GB_PARAMS].x); m.p1 = 0u; m.p2 = 0u;
            const float rough = LM_P_ROUGHNESS(m);
            normalRoughness[li] = make_uint2(lm_f32_to_f16(b.x) | (lm_f32_to_f16(b.y) << 16), lm_f32_to_f16(b.z) | (lm_f32_to_f16(rough) << 16));
        }
    }
}

// The merged radiance as the reference stores it: its pixel buffers are half4 surfaces (GPUMergeOutputChannels.cu:5-88 works on
// half4Ushort4; Half4.h:9-96 converts with __float2half = round to nearest even).  One rounding of the fp32 result — the report
// SURVEY.md §8 c6 asks for beside the fp32 contract (the reference's own chain of fp16 adds is order-dependent and racy, F9).
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_export_half4)(const float4* __restrict__ src, uint2* __restrict__ dst, uint32_t n)
{
    const uint32_t stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const float4 c = src[i];
        dst[i] = make_uint2(lm_f32_to_f16(c.x) | (lm_f32_to_f16(c.y) << 16), lm_f32_to_f16(c.z) | (lm_f32_to_f16(c.w) << 16));
    }
}

// Multi-GPU tiles: a w x h rectangle of RGBA32F pixels between two pitched device images (a rank's tile out of its render window into the gather's send buffer; a
// gathered tile into the assembled frame).  One float4 per lane, rows contiguous: a plain HBM stream.  Kept in this module so that the per-frame path of the tiled
// renderer launches nothing but kernels that are already resident (LOG.md round 5 item 1: a lazily loaded PyTorch copy kernel faulted at its first launch there).
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_copy_rect)(float4* __restrict__ dst, uint32_t dstPitch, const float4* __restrict__ src, uint32_t srcPitch, uint32_t w, uint32_t h)
{
    const uint32_t n = w * h, stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t y = i / w, x = i - y * w;
        dst[(size_t)y * dstPitch + x] = src[(size_t)y * srcPitch + x];
    }
}

// Multi-GPU seams: a single GPU advances the reservoir swap chain once per wave that holds a ray ANYWHERE in the image.  A rank
// only sees its window, so the ranks exchange the number of waves they executed (export), take the maximum (all-reduce) and
// advance by the difference (import).
extern "C" __global__ void KN(lm_k_wave_sync)(int* swap, int* io, int import)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (!import) { io[0] = swap[1]; return; }
    const int global = io[0] > swap[1] ? io[0] : swap[1];
    swap[0] = (swap[0] + (global - swap[1])) & 1;
    swap[1] = global;
}

// Multi-GPU seams: pack / unpack a rectangle (window-local pixels) of the reservoirs the NEXT frame's temporal pass reads as
// "previous" (5 float4 per pixel: the 64-byte hot record + the contribution plane).  Runs after the merge of a frame, which has
// advanced the swap chain, so LM_RES_PREV already names that buffer.  A rank sends the part of its tile that lies in a
// neighbour's halo and receives its own halo ring the same way (lumenrenderer_amd/tiles.py exchange_history).
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_history_copy)(LmFrame fr, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float4* __restrict__ buf, int import)
{
    const int rp = lm_res_idx(fr, LM_RES_PREV);
    float4* hot = fr.res[rp];
    float4* con = fr.resC[rp];
    if (import && blockIdx.x == 0 && threadIdx.x == 0) fr.swap[2 + rp] = 1;      // the next temporal pass must read what arrives here
    const uint32_t n = w * h, stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t yy = i / w, xx = i - yy * w;
        const uint32_t li = (y0 + yy) * fr.ww + (x0 + xx);
        float4* b = buf + 5u * i;
        if (import) { for (uint32_t k = 0; k < 4u; k++) hot[4u * li + k] = b[k]; con[li] = b[4]; }
        else { for (uint32_t k = 0; k < 4u; k++) b[k] = hot[4u * li + k]; b[4] = con[li]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BVH refit for moved instances (reference: the per-frame instance acceleration-structure rebuild of PTScene.cpp:74-156,
// PTMeshInstance.cpp:123-178).  Topology and leaf contents stay; triangles are re-transformed, Woop packets recomputed
// (bit-identical to the host builder: lm_tri.h), boxes propagated bottom-up level by level and re-quantised against the
// new scene box.  Boxes only cull, so the hit records equal those of a freshly built tree.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lm_ordered(float f) { const uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lm_unordered(uint32_t u) { return u2f((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ float lm_wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float lm_wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
// bounds[0..2] min, [3..5] max (order-preserving encoding), [6] max |coordinate| (bits of a non-negative float).
// The seven bounds sit in one cache line, and atomics on one line retire at ~ 88 per microsecond: one set per WAVEFRONT (rounds 1 - 5: 4 098 wavefronts x 7 for C2) was
// 326 of the kernel's 333 us.  Now a grid of at most two blocks per CU loops over the slots, a block reduces its bounds through LDS and issues ONE set of atomics
// (min / max are order independent: the same bits).
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_refit_tris)(LmScene sc, uint32_t nSlots, float4* __restrict__ triBox, uint32_t* bounds)
{
    __shared__ float s_red[7 * (LM_BLOCK / 64)];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, maxAbs = 0.f;
    for (uint32_t s = blockIdx.x * LM_BLOCK + threadIdx.x; s < nSlots; s += gridDim.x * LM_BLOCK) {
        const uint2 id = sc.triId[s];
        const LmEntry e = sc.entries[id.x];
        float tri[9];
        for (int k = 0; k < 3; k++) {
            const uint32_t vi = sc.indices[e.idxBase + 3u * id.y + (uint32_t)k];
            const float4 p = sc.verts[3u * (e.vertBase + vi)];
            // rows 0..2 of the world matrix, operation order of the host flatten (sutil Matrix4x4 * float4)
            tri[3 * k + 0] = e.m[0] * p.x + e.m[1] * p.y + e.m[2] * p.z + e.m[3] * 1.f;
            tri[3 * k + 1] = e.m[4] * p.x + e.m[5] * p.y + e.m[6] * p.z + e.m[7] * 1.f;
            tri[3 * k + 2] = e.m[8] * p.x + e.m[9] * p.y + e.m[10] * p.z + e.m[11] * 1.f;
        }
        sc.packets[s] = lm_make_packet(tri);
        float tlo[3] = {INFINITY, INFINITY, INFINITY}, thi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 9; k++) { tlo[k % 3] = fminf(tlo[k % 3], tri[k]); thi[k % 3] = fmaxf(thi[k % 3], tri[k]); maxAbs = fmaxf(maxAbs, fabsf(tri[k])); }
        triBox[2u * s] = make_float4(tlo[0], tlo[1], tlo[2], 0.f);
        triBox[2u * s + 1u] = make_float4(thi[0], thi[1], thi[2], 0.f);
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], tlo[k]); hi[k] = fmaxf(hi[k], thi[k]); }
    }
    for (int k = 0; k < 3; k++) { lo[k] = lm_wave_min(lo[k]); hi[k] = lm_wave_max(hi[k]); }
    maxAbs = lm_wave_max(maxAbs);
    const uint32_t wave = threadIdx.x >> 6;
    if (lm_lane() == 0u) { for (int k = 0; k < 3; k++) { s_red[7u * wave + k] = lo[k]; s_red[7u * wave + 3 + k] = hi[k]; } s_red[7u * wave + 6] = maxAbs; }
    __syncthreads();
    if (threadIdx.x < 7u) {
        const uint32_t k = threadIdx.x;
        float v = s_red[k];
        for (uint32_t w = 1; w < LM_BLOCK / 64; w++) v = k < 3u ? fminf(v, s_red[7u * w + k]) : fmaxf(v, s_red[7u * w + k]);
        // an empty block (no slot) holds +inf / -inf / 0: harmless for min / max, skipped anyway
        if (k < 3u) { if (v != INFINITY) atomicMin(bounds + k, lm_ordered(v)); }
        else if (k < 6u) { if (v != -INFINITY) atomicMax(bounds + k, lm_ordered(v)); }
        else atomicMax(bounds + 6, f2u(v));
    }
}
// scene box -> quantisation frame (the formulas of lm_build_bvh), then re-arm the bounds for the next refit
extern "C" __global__ void KN(lm_k_refit_quant)(uint32_t* bounds, float* quant)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float pad = u2f(bounds[6]) * (1.0f / 32768.0f);
    for (int k = 0; k < 3; k++) {
        float smin = lm_unordered(bounds[k]), smax = lm_unordered(bounds[3 + k]);
        if (!(smin <= smax)) { smin = 0.f; smax = 1.f; }
        const float m = 4.f * pad + 1e-6f * fmaxf(fabsf(smin), fabsf(smax)) + 1e-30f;
        smin -= m; smax += m;
        quant[k] = smin;
        quant[3 + k] = (smax - smin) / 65535.0f;
        bounds[k] = 0xffffffffu; bounds[3 + k] = 0u;
    }
    quant[6] = pad;
    bounds[6] = 0u;
}
__device__ __forceinline__ uint32_t lm_quant_axis(float lo, float hi, float qmin, float qstep)
{
    const double inv = 1.0 / (double)qstep;
    long long ql = (long long)floor(((double)lo - (double)qmin) * inv) - LM_QUANT_MARGIN;
    long long qh = (long long)ceil(((double)hi - (double)qmin) * inv) + LM_QUANT_MARGIN;
    ql = ql < 0 ? 0 : (ql > 65535 ? 65535 : ql); qh = qh < 0 ? 0 : (qh > 65535 ? 65535 : qh);
    return (uint32_t)ql | ((uint32_t)qh << 16);
}
// one depth level of the 4-wide tree (children of these nodes are leaves or nodes of deeper, already refitted levels)
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_refit_level)(LmScene sc, const uint32_t* __restrict__ levelNodes, uint32_t count, const float4* __restrict__ triBox, float4* nodeBox)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t n = levelNodes[i];
    const float pad = sc.quant[6];
    lf3 nlo = v3(INFINITY), nhi = v3(-INFINITY);
    for (int k = 0; k < LM_WIDTH; k++) {
        uint4 c = sc.nodes[n].c[k];
        const int ref = (int)c.w;
        if (ref == LM_REF_NONE) continue;
        lf3 lo = v3(INFINITY), hi = v3(-INFINITY);
        if (ref < 0) {
            const uint32_t leaf = (uint32_t)(~ref), first = leaf >> 3, cnt = (leaf & 7u) + 1u;
            for (uint32_t t = 0; t < cnt; t++) {
                const float4 a = triBox[2u * (first + t)], b = triBox[2u * (first + t) + 1u];
                lo = v3(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z)); hi = v3(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z));
            }
        } else {
            const float4 a = nodeBox[2u * (uint32_t)ref], b = nodeBox[2u * (uint32_t)ref + 1u];
            lo = v3(a); hi = v3(b);
        }
        nlo = v3(fminf(nlo.x, lo.x), fminf(nlo.y, lo.y), fminf(nlo.z, lo.z)); nhi = v3(fmaxf(nhi.x, hi.x), fmaxf(nhi.y, hi.y), fmaxf(nhi.z, hi.z));
        c.x = lm_quant_axis(lo.x - pad, hi.x + pad, sc.quant[0], sc.quant[3]);
        c.y = lm_quant_axis(lo.y - pad, hi.y + pad, sc.quant[1], sc.quant[4]);
        c.z = lm_quant_axis(lo.z - pad, hi.z + pad, sc.quant[2], sc.quant[5]);
        sc.nodes[n].c[k] = c;
    }
    nodeBox[2u * n] = v4(nlo, 0.f);
    nodeBox[2u * n + 1u] = v4(nhi, 0.f);
}

// Ray reorder between waves (tuning key "sort_rays"): a counting sort of a continuation-ray queue by (origin cell, direction octant), so that
// the 64 rays a wavefront traces together start in the same part of the scene and head the same way.  Three launches: count per bin
// (4 096 bins = 8 x 8 x 8 cells of the scene box x 8 octants, non-returning atomics), exclusive scan, scatter (one returning atomic per
// ray on its bin's cursor).  The order inside a bin is whatever the atomics give — every consumer finds its pixel in rayD.w, so results
// do not depend on it (test_schedules_do_not_change_results).
#define LM_SORT_BINS 4096u
__device__ __forceinline__ uint32_t lm_sort_key(const LmScene& sc, const float4& o, const float4& d)
{
    uint32_t key = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    const float c[3] = {o.x, o.y, o.z};
    for (int k = 0; k < 3; k++) {
        const float rel = (c[k] - sc.quant[k]) / (sc.quant[3 + k] * 65535.0f);
        const int cell = min(7, max(0, (int)(rel * 8.0f)));
        key |= (uint32_t)cell << (3 + 3 * k);
    }
    return key;
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_sort_count)(LmScene sc, const float4* __restrict__ rayO, const float4* __restrict__ rayD, const uint32_t* __restrict__ countPtr, uint32_t* bins)
{
    const uint32_t n = *countPtr, stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) atomicAdd(bins + lm_sort_key(sc, rayO[i], rayD[i]), 1u);
}
// bins -> exclusive prefix in cursors; bins are zeroed for the next use.  One block of 1024 threads, 4 bins each.
extern "C" __global__ void __launch_bounds__(1024)
KN(lm_k_sort_scan)(uint32_t* bins, uint32_t* cursors)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x;
    uint32_t v[4], sum = 0;
    for (int k = 0; k < 4; k++) { v[k] = bins[4u * t + k]; bins[4u * t + k] = 0u; sum += v[k]; }
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t add = t >= off ? s_part[t - off] : 0u;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    uint32_t base = s_part[t] - sum;
    for (int k = 0; k < 4; k++) { cursors[4u * t + k] = base; base += v[k]; }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_sort_scatter)(LmScene sc, const float4* __restrict__ srcO, const float4* __restrict__ srcD, const float4* __restrict__ srcC,
                      float4* __restrict__ dstO, float4* __restrict__ dstD, float4* __restrict__ dstC, const uint32_t* __restrict__ countPtr, uint32_t* cursors)
{
    const uint32_t n = *countPtr, stride = gridDim.x * LM_BLOCK;
    for (uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x; i < n; i += stride) {
        const float4 o = srcO[i], d = srcD[i], c = srcC[i];
        const uint32_t pos = atomicAdd(cursors + lm_sort_key(sc, o, d), 1u);
        dstO[pos] = o; dstD[pos] = d; dstC[pos] = c;
    }
}

// Top-of-tree table for the queue traversal kernels (lm_layout.h LM_TOP_NODES): the first LM_TOP_NODES inner nodes in breadth-first
// order, with child references rewritten to table slots where the child made it into the table.  One wavefront, level by level
// (slots are handed out in lane order: deterministic); run whenever the node array changes (build, refit, assembly).
extern "C" __global__ void KN(lm_k_build_top)(const LmNodeW* __restrict__ nodes, LmNodeW* __restrict__ top)
{
#if LM_TOP_NODES
    __shared__ int s_src[LM_TOP_NODES];
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x != 0 || lane >= 64u) return;
    if (lane == 0u) s_src[0] = 0;
    __syncthreads();
    uint32_t begin = 0, count = 1;
    while (begin < count) {
        const uint32_t end = count;
        for (uint32_t base = begin; base < end; base += 64u) {
            const uint32_t s = base + lane;
            const bool valid = s < end;
            LmNodeW nd;
            uint32_t inner = 0;
            if (valid) {
                nd = nodes[s_src[s]];
                for (int j = 0; j < LM_WIDTH; j++) inner += ((int)nd.c[j].w >= 0 && (int)nd.c[j].w != LM_REF_NONE) ? 1u : 0u;
            }
            uint32_t prefix = inner;                                      // inclusive scan over the wavefront
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(prefix, o); if ((int)lane >= o) prefix += v; }
            const uint32_t total = __shfl(prefix, 63);
            uint32_t slot = count + prefix - inner;
            if (valid) {
                for (int j = 0; j < LM_WIDTH; j++) {
                    const int ref = (int)nd.c[j].w;
                    if (ref < 0 || ref == LM_REF_NONE) continue;
                    if (slot < (uint32_t)LM_TOP_NODES) { s_src[slot] = ref; nd.c[j].w = (uint32_t)(LM_TOP_BASE + (int)slot); }
                    slot++;
                }
                top[s] = nd;
            }
            count = min((uint32_t)LM_TOP_NODES, count + total);
            __syncthreads();
        }
        begin = end;
    }
#endif
}

// Schedule fuzzing (tuning key "fuzz"): a single wavefront that holds its stream busy for a while.  Inserted at random in front of
// the launches of a frame, it shifts what overlaps with what; a missing dependency between streams then shows as a changed image.
extern "C" __global__ void KN(lm_k_spin)(uint32_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)ticks) __builtin_amdgcn_s_sleep(8);
}

#define LM_GRID(g) dim3((unsigned)(g)), dim3(LM_BLOCK), 0, s
// Residency caps: a launch may carry unused dynamic LDS so that fewer blocks of a kernel fit a CU (160 KB of LDS per CU: 40 960 B = four blocks, 54 272 B = three).
// A kernel that saturates one unit (vector ALUs, the L1's miss queue, HBM) with fewer waves than its registers allow only takes room from the other streams' kernels
// by holding more.  Defaults below are measured (profiles/r03_residency_caps_ab.txt); LUMEN_MI_CAP_<NAME>=bytes overrides one for experiments.
#if !LM_INSTRUMENT
#include <cstdlib>
static unsigned lm_cap(const char* name, unsigned dflt) { const char* e = getenv(name); return e ? (unsigned)atoi(e) : dflt; }
#define LM_CAP(NAME, DFLT) ([]() { static const unsigned v = lm_cap("LUMEN_MI_CAP_" NAME, DFLT); return v; }())
#else
#define LM_CAP(NAME, DFLT) 0u
#endif
#define LM_GRID_CAP(g, NAME, DFLT) dim3((unsigned)(g)), dim3(LM_BLOCK), LM_CAP(NAME, DFLT), s

static void l_primary(hipStream_t s, int g, LmFrame fr, LmCamera cam, uint32_t frameCount) { hipLaunchKernelGGL(KN(lm_k_primary), LM_GRID_CAP(g, "PRIMARY", 0u), fr, cam, frameCount); }
static void l_trace_primary(hipStream_t s, int g, LmScene sc, LmFrame fr, LmCamera cam, uint32_t frameCount, uint4* hits, float tmin, float tmax)
{ hipLaunchKernelGGL(KN(lm_k_trace_primary_packet), LM_GRID(g), sc, fr, cam, frameCount, hits, tmin, tmax); }
static void l_trace_closest(hipStream_t s, int g, LmScene sc, const float4* o, const float4* d, const uint32_t* cnt, uint4* hits, float tmin, float tmax, uint32_t* counters, int refillBelow, const float* eye)
{
    const float4 e = eye ? make_float4(eye[0], eye[1], eye[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
#if !LM_INSTRUMENT
    // refillBelow < 0: the queue holds coherent bundles (primary rays): packet traversal.  (The counting build always takes the per-lane
    // kernel: its node / triangle counts are per ray.)
    if (refillBelow < 0) { hipLaunchKernelGGL(KN(lm_k_trace_closest_packet), LM_GRID(g), sc, o, d, cnt, hits, tmin, tmax, e); return; }
#endif
    hipLaunchKernelGGL(KN(lm_k_trace_closest), LM_GRID(g), sc, o, d, cnt, hits, tmin, tmax, counters, refillBelow < 0 ? 0 : refillBelow, e);
}
static void l_extract0(hipStream_t s, int g, LmScene sc, LmFrame fr, LmCamera cam, int cur, uint32_t seed2, int doIndirect, int outQ, uint32_t* outCount)
{ hipLaunchKernelGGL(KN(lm_k_extract0), LM_GRID_CAP(g, "EXTRACT", 0u), sc, fr, cam, cur, seed2, doIndirect, outQ, outCount); }
static void l_shade_wave(hipStream_t s, int g, LmScene sc, LmFrame fr, int inQ, const uint32_t* inCount, uint32_t seed, uint32_t seed2, int doIndirect, uint32_t* outCount, uint32_t* shadowCount)
{
    // bit 1 of doIndirect / of inQ (path tail): tuning key fast_shade, the NEE contribution in the fast arithmetic policy
    if (doIndirect & 2) hipLaunchKernelGGL(KN(lm_k_shade_wave_fs), LM_GRID(g), sc, fr, inQ, inCount, seed, seed2, doIndirect & 1, outCount, shadowCount);
    else hipLaunchKernelGGL(KN(lm_k_shade_wave), LM_GRID(g), sc, fr, inQ, inCount, seed, seed2, doIndirect, outCount, shadowCount);
}
static void l_path_tail(hipStream_t s, int g, LmScene sc, LmFrame fr, int inQ, const uint32_t* inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave)
{
    const bool fs = (inQ & 2) != 0; inQ &= 1;
    if (lanesPerWave >= 1000) {      // tuning key tail_repack: 256 paths per block, repacked after every depth
        if (fs) hipLaunchKernelGGL(KN(lm_k_path_tail_repack_fs), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0);
        else hipLaunchKernelGGL(KN(lm_k_path_tail_repack), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0);
        return;
    }
    if (lanesPerWave < 0) { if (fs) hipLaunchKernelGGL(KN(lm_k_path_tail_pair_fs), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave);
                            else hipLaunchKernelGGL(KN(lm_k_path_tail_pair), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave); }
    else if (fs) hipLaunchKernelGGL(KN(lm_k_path_tail_fs), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave);
    else hipLaunchKernelGGL(KN(lm_k_path_tail), LM_GRID(g), sc, fr, inQ, inCount, depth0, depthMax, seed0, lanesPerWave);
}
static void l_trace_shadow(hipStream_t s, int g, LmScene sc, LmFrame fr, const uint32_t* cnt, float tmin, int refillBelow) { hipLaunchKernelGGL(KN(lm_k_trace_shadow), LM_GRID(g), sc, fr, cnt, tmin, refillBelow); }
static void l_fill_bags(hipStream_t s, LmScene sc, LmFrame fr, uint32_t seed, uint32_t total) { hipLaunchKernelGGL(KN(lm_k_fill_bags), LM_GRID((total + LM_BLOCK - 1) / LM_BLOCK), sc, fr, seed, total); }
static void l_pick_primary(hipStream_t s, int tiles, LmScene sc, LmFrame fr, int cur, int rc, uint32_t seed, uint32_t* visCount, int fast)
{
    const int wideSel = (fast >> 6) & 3;          // tuning key "pick_wide" + 1 (frame.cpp): 0 unset = 1, else 1 never, 2 fast mode only (default), 3 both modes
    fast &= 3;
    const bool ldsLights = sc.numLights <= LM_PICK_LDS_LIGHTS && LM_PICK_LDS_LIGHTS > 0u;
    const bool ldsBig = !ldsLights && LM_PICK_STATIC_LDS && sc.numLights <= LM_PICK_LDS_LIGHTS_BIG && LM_PICK_LDS_LIGHTS_BIG > LM_PICK_LDS_LIGHTS;      // the 32-KB table
    const size_t lightBytes = LM_PICK_STATIC_LDS ? 0u : (size_t)64 * sc.numLights + 16u;         // dynamic LDS of the *_lds kernels
    // The exact instantiation (106 registers, long and uneven tiles) LOSES with the big block: 1 366 -> 1 569 us alone on C3, frame -5 %; the fast one gains (762 -> 519 us,
    // lazy frame +7.9 %): profiles/r06_pick_wide.txt.  So the wide block is the fast mode's; the exact kernel stays selectable for the parity test that runs it against the oracle.
    const bool wideOn = wideSel == 3 || (wideSel != 1 && fast);
    // (shorter lists keep the 256-thread blocks: the wide block at LowpolyRoom's 414 lights -0.6 %, at C2's two lights -8.6 % — more resident waves of a kernel that saturates the vector ALUs)
    const bool wide = wideOn && !ldsLights && !ldsBig && sc.numLights <= LM_PICK_WIDE_LIGHTS && LM_PICK_WIDE_LIGHTS > 0u;      // four tiles per block around one table
    if (wide) {
        static bool allowed[64] = {};     // more than 64 KB of dynamic LDS has to be asked for, once per kernel and device
        int dev = 0; (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !allowed[dev]) {
            (void)hipFuncSetAttribute((const void*)KN(lm_k_pick_primary_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64u * LM_PICK_WIDE_LIGHTS + 16u));
            (void)hipFuncSetAttribute((const void*)KN(lm_k_pick_primary_fast_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64u * LM_PICK_WIDE_LIGHTS + 16u));
            if (dev >= 0 && dev < 64) allowed[dev] = true;
        }
        const dim3 grid((unsigned)(tiles + 3) / 4u), block(4 * LM_BLOCK);
        const size_t bytes = (size_t)64 * sc.numLights + 16u;
        if (fast) {
            hipLaunchKernelGGL(KN(lm_k_pick_primary_fast_wide), grid, block, bytes, s, sc, fr, cur, rc, seed, visCount, (uint32_t)tiles);
            if (fast > 1) hipLaunchKernelGGL(KN(lm_k_pick_primary_rare), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
        } else hipLaunchKernelGGL(KN(lm_k_pick_primary_wide), grid, block, bytes, s, sc, fr, cur, rc, seed, visCount, (uint32_t)tiles);
        return;
    }
    if (fast) {
        if (ldsLights && LM_PICK_PERSIST > 0) {
            int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const unsigned grid = (unsigned)std::min(tiles, cus * (int)LM_PICK_PERSIST);
            hipLaunchKernelGGL(KN(lm_k_pick_primary_fast_lds_persist), dim3(grid), dim3(LM_BLOCK), (size_t)64 * sc.numLights + 16u, s, sc, fr, cur, rc, seed, visCount, (uint32_t)tiles);
        } else if (ldsLights) hipLaunchKernelGGL(KN(lm_k_pick_primary_fast_lds), dim3((unsigned)tiles), dim3(LM_BLOCK), lightBytes, s, sc, fr, cur, rc, seed, visCount);
        else if (ldsBig) hipLaunchKernelGGL(KN(lm_k_pick_primary_fast_lds_big), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
        else hipLaunchKernelGGL(KN(lm_k_pick_primary_fast), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
        if (fast > 1) hipLaunchKernelGGL(KN(lm_k_pick_primary_rare), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
    } else if (ldsLights) hipLaunchKernelGGL(KN(lm_k_pick_primary_lds), dim3((unsigned)tiles), dim3(LM_BLOCK), lightBytes, s, sc, fr, cur, rc, seed, visCount);
    else if (ldsBig) hipLaunchKernelGGL(KN(lm_k_pick_primary_lds_big), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
    else hipLaunchKernelGGL(KN(lm_k_pick_primary), LM_GRID(tiles), sc, fr, cur, rc, seed, visCount);
}
static void l_trace_shade(hipStream_t s, int g, LmScene sc, LmFrame fr, int rc, const uint32_t* cnt, int refillBelow, int pass)
{
#if !LM_INSTRUMENT
    if (refillBelow < 0) { hipLaunchKernelGGL(KN(lm_k_restir_trace_shade_packet), LM_GRID(g), sc, fr, rc, cnt, pass); return; }      // coherent queue: packet traversal
#endif
    hipLaunchKernelGGL(KN(lm_k_restir_trace_shade), LM_GRID(g), sc, fr, rc, cnt, refillBelow < 0 ? 0 : refillBelow, pass);
}
static void l_temporal(hipStream_t s, int g, LmFrame fr, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount, int fast)
{ if (fast) { hipLaunchKernelGGL(KN(lm_k_restir_temporal_fast), LM_GRID_CAP(g, "TEMPORAL", 0u), fr, cur, prev, rc, rp, rf, seed, visCount); if (fast > 1) hipLaunchKernelGGL(KN(lm_k_restir_temporal_rare), LM_GRID(g), fr, cur, prev, rc, rp, rf, seed, visCount); } else hipLaunchKernelGGL(KN(lm_k_restir_temporal), LM_GRID(g), fr, cur, prev, rc, rp, rf, seed, visCount); }
static void l_spatial(hipStream_t s, int g, LmFrame fr, int cur, int rin, int rout, uint32_t seed, int margin, int pass, int fast)
{
    // fast | 16: the first pass stages its probe window in LDS (32 x 32 pixel tiles, 1024-thread blocks; tuning key spatial_lds)
    if ((fast & 32) && pass == 0) {
        hipLaunchKernelGGL(KN(lm_k_restir_spatial_fast_lds16), LM_GRID(g), fr, cur, rin, rout, seed, margin);
        if ((fast & 15) > 1) hipLaunchKernelGGL(KN(lm_k_restir_spatial_rare), LM_GRID(g), fr, cur, rin, rout, seed, margin, pass);
        return;
    }
    if ((fast & 16) && pass == 0) {
        const int g32 = (int)(((fr.ww + 31u) / 32u) * ((fr.wh + 31u) / 32u));
        hipLaunchKernelGGL(KN(lm_k_restir_spatial_fast_lds), dim3((unsigned)g32), dim3(1024), 0, s, fr, cur, rin, rout, seed, margin);
        if ((fast & 15) > 1) hipLaunchKernelGGL(KN(lm_k_restir_spatial_rare), LM_GRID(g), fr, cur, rin, rout, seed, margin, pass);
        return;
    }
    if ((fast & 64) && pass == 1 && !fr.deferred && (fast & 15) <= 1) {      // second pass + combine (fr.fuseRc, fr.fuseSeed)
        if (fast & 15) hipLaunchKernelGGL(KN(lm_k_restir_spatial_fast_fused), LM_GRID_CAP(g, "SPATIAL", 0u), fr, cur, rin, rout, seed);
        else hipLaunchKernelGGL(KN(lm_k_restir_spatial_fused), LM_GRID(g), fr, cur, rin, rout, seed);
        return;
    }
    fast &= 15;
    if (fr.deferred) {        // the previous frame's pass (lazy reuse): looping grid, returns at once unless owed
        const int gd = g < 2048 ? g : 2048;
        if (fast) { hipLaunchKernelGGL(KN(lm_k_restir_spatial_fast_deferred), LM_GRID(gd), fr, cur, rin, rout, seed, margin, pass, g); if (fast > 1) hipLaunchKernelGGL(KN(lm_k_restir_spatial_rare_deferred), LM_GRID(gd), fr, cur, rin, rout, seed, margin, pass, g); }
        else hipLaunchKernelGGL(KN(lm_k_restir_spatial_deferred), LM_GRID(gd), fr, cur, rin, rout, seed, margin, pass, g);
        return;
    }
    if (fast) { hipLaunchKernelGGL(KN(lm_k_restir_spatial_fast), LM_GRID_CAP(g, "SPATIAL", 0u), fr, cur, rin, rout, seed, margin, pass); if (fast > 1) hipLaunchKernelGGL(KN(lm_k_restir_spatial_rare), LM_GRID(g), fr, cur, rin, rout, seed, margin, pass); } else hipLaunchKernelGGL(KN(lm_k_restir_spatial), LM_GRID(g), fr, cur, rin, rout, seed, margin, pass);
}
static void l_combine(hipStream_t s, int g, LmFrame fr, int cur, int rc, int rs, uint32_t seed, int fast)
{
    if (fr.deferred) {
        const int gd = g < 2048 ? g : 2048;
        if (fast) { hipLaunchKernelGGL(KN(lm_k_restir_combine_fast_deferred), LM_GRID(gd), fr, cur, rc, rs, seed, g); if (fast > 1) hipLaunchKernelGGL(KN(lm_k_restir_combine_rare_deferred), LM_GRID(gd), fr, cur, rc, rs, seed, g); }
        else hipLaunchKernelGGL(KN(lm_k_restir_combine_deferred), LM_GRID(gd), fr, cur, rc, rs, seed, g);
        return;
    }
  if (fast) { hipLaunchKernelGGL(KN(lm_k_restir_combine_fast), LM_GRID_CAP(g, "COMBINE", 0u), fr, cur, rc, rs, seed); if (fast > 1) hipLaunchKernelGGL(KN(lm_k_restir_combine_rare), LM_GRID(g), fr, cur, rc, rs, seed); } else hipLaunchKernelGGL(KN(lm_k_restir_combine), LM_GRID(g), fr, cur, rc, rs, seed); }
static void l_clear(hipStream_t s, int g, float4* p, uint32_t n) { hipLaunchKernelGGL(KN(lm_k_clear_f4), LM_GRID(g), p, n); }
static void l_merge(hipStream_t s, int g, LmFrame fr, int blend, uint32_t blendCount, int depthMax) { hipLaunchKernelGGL(KN(lm_k_merge_output), LM_GRID_CAP(g, "MERGE", 0u), fr, blend, blendCount, depthMax); }
static void l_query_any(hipStream_t s, int g, LmScene sc, const float4* o, const float4* d, uint32_t n, float tmin, uint32_t* occ, uint32_t* counters) { hipLaunchKernelGGL(KN(lm_k_query_any), LM_GRID(g), sc, o, d, n, tmin, occ, counters); }
static void l_query_closest(hipStream_t s, int g, LmScene sc, const float4* o, const float4* d, uint32_t n, float tmin, float tmax, uint4* id, float4* uvt, uint32_t* counters)
{ hipLaunchKernelGGL(KN(lm_k_query_closest_raw), LM_GRID(g), sc, o, d, n, tmin, tmax, id, uvt, counters); }
static void l_export_aux(hipStream_t s, int g, LmFrame fr, int cur, float minD, float maxD, float* depth, uint2* nr) { hipLaunchKernelGGL(KN(lm_k_export_aux), LM_GRID(g), fr, cur, minD, maxD, depth, nr); }
static void l_refit_tris(hipStream_t s, LmScene sc, uint32_t nSlots, float4* triBox, uint32_t* bounds)
{ const uint32_t g = std::max(1u, std::min((nSlots + LM_BLOCK - 1) / LM_BLOCK, 512u)); hipLaunchKernelGGL(KN(lm_k_refit_tris), LM_GRID(g), sc, nSlots, triBox, bounds); }      // <= two blocks per CU: one set of bound atomics per block
static void l_refit_quant(hipStream_t s, uint32_t* bounds, float* quant) { hipLaunchKernelGGL(KN(lm_k_refit_quant), dim3(1), dim3(64), 0, s, bounds, quant); }
static void l_refit_level(hipStream_t s, LmScene sc, const uint32_t* levelNodes, uint32_t count, const float4* triBox, float4* nodeBox) { hipLaunchKernelGGL(KN(lm_k_refit_level), LM_GRID((count + LM_BLOCK - 1) / LM_BLOCK), sc, levelNodes, count, triBox, nodeBox); }
#if LUMEN_MI_TEST_HOOKS
#define LM_HOOKS_PART 2
#include "lm_hooks.h"
#else      // a library without its test surface: the table's hook entries are null (nothing in the frame path calls them)
#define l_test_bsdf nullptr
#define l_test_math nullptr
#define l_test_restir nullptr
#define l_kat_pack_surfaces nullptr
#define l_kat_reservoirs nullptr
#define l_kat_resolve nullptr
#define l_kat_shade nullptr
#define l_kat_extract nullptr
#define l_kat_tex2d nullptr
#endif
static void l_history_copy(hipStream_t s, int g, LmFrame fr, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float4* buf, int import)
{ hipLaunchKernelGGL(KN(lm_k_history_copy), LM_GRID(g), fr, x0, y0, w, h, buf, import); }
static void l_reuse_counts(hipStream_t s, LmFrame fo, int was, const uint32_t* list, const uint32_t* listCount, uint32_t seed)
{ for (int phase = 0; phase < 2; phase++) hipLaunchKernelGGL(KN(lm_k_reuse_counts), dim3(64), dim3(LM_BLOCK), 0, s, fo, was, list, listCount, seed, phase); }
static void l_reuse_settle(hipStream_t s, LmFrame fr) { hipLaunchKernelGGL(KN(lm_k_reuse_settle), dim3(1), dim3(64), 0, s, fr); }
static void l_wave_sync(hipStream_t s, int* swap, int* io, int import) { hipLaunchKernelGGL(KN(lm_k_wave_sync), dim3(1), dim3(64), 0, s, swap, io, import); }
static void l_sort_rays(hipStream_t s, int g, LmScene sc, const float4* srcO, const float4* srcD, const float4* srcC, float4* dstO, float4* dstD, float4* dstC,
                        const uint32_t* cnt, uint32_t* bins)
{
    hipLaunchKernelGGL(KN(lm_k_sort_count), LM_GRID(g), sc, srcO, srcD, cnt, bins);
    hipLaunchKernelGGL(KN(lm_k_sort_scan), dim3(1), dim3(1024), 0, s, bins, bins + LM_SORT_BINS);
    hipLaunchKernelGGL(KN(lm_k_sort_scatter), LM_GRID(g), sc, srcO, srcD, srcC, dstO, dstD, dstC, cnt, bins + LM_SORT_BINS);
}
static void l_export_half4(hipStream_t s, int g, const float4* src, uint2* dst, uint32_t n) { hipLaunchKernelGGL(KN(lm_k_export_half4), LM_GRID(g), src, dst, n); }
static void l_copy_rect(hipStream_t s, int g, float4* dst, uint32_t dp, const float4* src, uint32_t sp, uint32_t w, uint32_t h) { hipLaunchKernelGGL(KN(lm_k_copy_rect), LM_GRID(g), dst, dp, src, sp, w, h); }
static void l_build_top(hipStream_t s, const LmNodeW* nodes, LmNodeW* top) { hipLaunchKernelGGL(KN(lm_k_build_top), dim3(1), dim3(64), 0, s, nodes, top); }
static void l_spin(hipStream_t s, uint32_t ticks) { hipLaunchKernelGGL(KN(lm_k_spin), dim3(1), dim3(64), 0, s, ticks); }

#if LM_INSTRUMENT
extern "C" const LmKernelTable* lm_kernel_table_instrumented()
#elif LM_NOSLP_VARIANT
extern "C" const LmKernelTable* lm_kernel_table_noslp()
#else
extern "C" const LmKernelTable* lm_kernel_table()
#endif
{
    static const LmKernelTable t = {l_primary, l_trace_closest, l_extract0, l_shade_wave, l_trace_shadow, l_path_tail, l_fill_bags, l_pick_primary,
                                    l_trace_shade, l_temporal, l_spatial, l_combine, l_clear, l_merge, l_query_any, l_query_closest, l_export_aux, l_refit_tris, l_refit_quant, l_refit_level, l_test_bsdf, l_test_math, l_spin, l_history_copy, l_wave_sync, l_test_restir, l_build_top, l_export_half4, l_sort_rays, l_reuse_settle, l_reuse_counts, l_trace_primary,
                                    l_kat_pack_surfaces, l_kat_reservoirs, l_kat_resolve, l_kat_shade, l_kat_extract, l_kat_tex2d, l_copy_rect};
    return &t;
}
