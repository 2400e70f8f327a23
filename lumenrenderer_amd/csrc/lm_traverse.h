// lm_traverse.h — device code: 4-wide BVH traversal (watertight ray / triangle test, hybrid LDS / global stack) and the queue
// traversal with per-lane ray replacement.  Included by kernels.hip only (one translation unit, compiled twice: LM_INSTRUMENT).
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// Traversal of the 4-wide BVH (bvh.cpp) with the watertight triangle test (lm_tri_test).  Per-lane stack: 16 entries in LDS, interleaved by lane
// (bank-conflict free), deeper entries spill to a per-thread global area.
// Closest hit: minimum t in (tmin, tmax); equal t -> lower global triangle index (order independent).
// ---------------------------------------------------------------------------------------------------------------------
struct LmHit { float t, u, v; uint32_t slot; };
typedef float lm_f2 __attribute__((ext_vector_type(2)));

// Per-lane traversal stack: the first LM_STACK_LDS entries live in LDS ([level][lane]: one bank per lane), deeper entries
// (rare) spill to a per-thread global array, so that the LDS footprint (16 KB per 256-thread block) does not cap occupancy.
typedef __attribute__((address_space(3))) int lm_lds_int;      // explicit LDS pointer: ds_read / ds_write, never flat accesses
typedef uint32_t lm_u4v __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a class and cannot be read through an address-space pointer)
typedef __attribute__((address_space(3))) lm_u4v lm_lds_u4;
__device__ __forceinline__ uint4 lm_lds_read4(const lm_lds_u4* p) { const lm_u4v v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
struct LmStack { lm_lds_int* lds; int* spill; };
// Top of the tree staged in LDS (queue kernels): every ray starts at the root and spends its first steps in the same few nodes, so
// those records are copied once per block and read with ds_read_b128 instead of a global load per lane and step.
__device__ __forceinline__ lm_lds_u4* lm_stage_top(uint4* s_top, const LmScene& sc)
{
#if LM_TOP_NODES
    for (uint32_t i = threadIdx.x; i < (uint32_t)(LM_WIDTH * LM_TOP_NODES); i += LM_BLOCK) s_top[i] = ((const uint4*)sc.top)[i];
    __syncthreads();
    return (lm_lds_u4*)s_top;
#else
    return nullptr;
#endif
}
#if LM_INSTRUMENT
__device__ unsigned long long g_lmPushes[2];       // counting build: [0] stack pushes, [1] of which went to the global spill area
#endif
__device__ __forceinline__ void lm_push(const LmStack& st, int& sp, int v)
{
#if LM_INSTRUMENT
    atomicAdd(&g_lmPushes[sp < LM_STACK_LDS ? 0 : 1], 1ull);
#endif
    if (sp < LM_STACK_LDS) st.lds[sp * LM_BLOCK] = v; else st.spill[sp - LM_STACK_LDS] = v;
    sp++;
}
__device__ __forceinline__ int lm_pop(const LmStack& st, int& sp)
{
    --sp;
    // the LDS slot is read unconditionally (clamped index) and the spill slot only under a branch: a select between the
    // two POINTERS would turn every pop into a flat load, which is slower than ds_read and waits on both counters
    int v = st.lds[min(sp, LM_STACK_LDS - 1) * LM_BLOCK];
    if (sp >= LM_STACK_LDS) v = st.spill[sp - LM_STACK_LDS];
    return v;
}
__device__ __forceinline__ LmStack lm_make_stack(int* s_stack, const LmScene& sc)
{
    LmStack st;
    st.lds = (lm_lds_int*)(s_stack + threadIdx.x);
    st.spill = sc.spill + (size_t)(blockIdx.x * LM_BLOCK + threadIdx.x) * (LM_STACK_DEPTH - LM_STACK_LDS);
    return st;
}

// ---------------------------------------------------------------------------------------------------------------------
// Watertight ray / triangle test: Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection", JCGT 2(1), 2013.  The reference's queries run on OptiX
// (OptixWrapper.cpp:46-131, WaveFrontShaders.cu:63-76), where no ray passes between two triangles that share an edge or a vertex; an affine
// unit-triangle packet rounded per triangle (rounds 1 - 5) cannot give that: tests/test_gpu_watertight.py found a quarter of the rays aimed at a shared
// edge escaping.  Here the triangle is translated to the ray origin, the axes are permuted so that kz is the ray's dominant one, x and y are sheared
// along z, and the three 2-D edge functions are evaluated.  Each 2-D point is a function of (vertex, ray) alone, so neighbours see the same point; the
// sign of fl(a b) - fl(c d) is never wrong, only possibly zero, and a zero is resolved exactly by the difference of the two products' rounding errors
// (one fma each; the paper goes to double there).  The 2-D inside test is therefore EXACT on the points it is given.
// GPU form: the permutation costs no instruction per triangle — a packet stores every vertex as x y z x y (lm_tri.h), the permutation is cyclic, so
// (v[kx], v[ky], v[kz]) are the three consecutive floats at float offset kx of the record: one 12-byte load per vertex at a per-ray offset.  The ray's
// origin is permuted once at setup, where 1 / d[kz] is already at hand from the slab test; x and y travel as register pairs (packed fp32 subtract,
// fma and multiply: 9 + 6 instructions for the three sheared vertices and the three edge functions).  Per-ray state: 7 registers in place of
// origin + direction.
// Operation order is part of the definition: the CPU checker (tri_hit, under oracle/) performs the same fp32 operations.
// ---------------------------------------------------------------------------------------------------------------------
struct LmRayTri { lm_f2 oxy; float oz; lm_f2 sxy; float sz; uint32_t row; };      // (o[kx], o[ky]), o[kz]; shear (Sx, Sy), Sz; byte offset 4 kx into a vertex record
struct LmV3 { float x, y, z; };                                         // one permuted vertex: 12 bytes at 4-byte alignment
__device__ __forceinline__ LmV3 lm_load_v3(const char* p) { LmV3 v; __builtin_memcpy(&v, p, 12); return v; }
__device__ __forceinline__ float lm_edge_exact(float a, float b, float c, float d)      // a b - c d when its rounded value is zero: the exact sign
{
    asm volatile("" ::: "memory");          // a rare path: kept a branch of its own (no speculation, no packing with its siblings: those cost six registers in every traversal kernel)
    const float p = a * b;
    const float ep = fmaf(a, b, -p);
    const float q = c * d;
    return ep - fmaf(c, d, -q);
}
// a, b, c = the three vertices as (v[kx], v[ky], v[kz]); true and (t, u, v) for a hit with tmin < t < tmax; u / v = barycentric weight of the second / third vertex
__device__ __forceinline__ bool lm_tri_test(const LmV3& a, const LmV3& b, const LmV3& c, const LmRayTri& q, float tmin, float tmax, float& t, float& u, float& v)
{
    const float az = a.z - q.oz, bz = b.z - q.oz, cz = c.z - q.oz;
    const lm_f2 nS = -q.sxy;
    const lm_f2 A = __builtin_elementwise_fma(nS, (lm_f2){az, az}, (lm_f2){a.x, a.y} - q.oxy);
    const lm_f2 B = __builtin_elementwise_fma(nS, (lm_f2){bz, bz}, (lm_f2){b.x, b.y} - q.oxy);
    const lm_f2 C = __builtin_elementwise_fma(nS, (lm_f2){cz, cz}, (lm_f2){c.x, c.y} - q.oxy);
    const lm_f2 pu = C * B.yx, pv = A * C.yx, pw = B * A.yx;                    // (Cx By, Cy Bx), (Ax Cy, Ay Cx), (Bx Ay, By Ax)
    float U, V, W;                              // as instructions: the optimiser would pair two of the three subtractions behind three register moves
    asm("v_sub_f32 %0, %1, %2" : "=v"(U) : "v"(pu.x), "v"(pu.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(V) : "v"(pv.x), "v"(pv.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(W) : "v"(pw.x), "v"(pw.y));
    if (fminf(fminf(U, V), W) < 0.f && fmaxf(fmaxf(U, V), W) > 0.f) return false;           // signs of non-zero edge functions are exact: outside
    if (U == 0.f || V == 0.f || W == 0.f) {                    // on an edge within fp32 (rare): exact signs for the zeros, then the inside test again
        if (U == 0.f) U = lm_edge_exact(C.x, B.y, C.y, B.x);
        if (V == 0.f) V = lm_edge_exact(A.x, C.y, A.y, C.x);
        if (W == 0.f) W = lm_edge_exact(B.x, A.y, B.y, A.x);
        if ((U < 0.f || V < 0.f || W < 0.f) && (U > 0.f || V > 0.f || W > 0.f)) return false;
    }
    const float det = U + V + W;
    if (det == 0.f) return false;
    const float T = fmaf(W, q.sz * cz, fmaf(V, q.sz * bz, U * (q.sz * az)));
    const float rdet = 1.0f / det;
    t = T * rdet;
    if (!(t > tmin && t < tmax)) return false;
    u = V * rdet; v = W * rdet;
    return true;
}
// The whole packet is fetched up front (three independent 12-byte loads from one cache line, one wait): a leaf visit costs one memory round trip per triangle.
__device__ __forceinline__ bool lm_tri(const LmTriPacket* __restrict__ packets, uint32_t slot, const LmRayTri& q,
                                       float tmin, float tmax, float& t, float& u, float& v)
{
    const char* base = (const char*)(packets + slot) + q.row;
    const LmV3 a = lm_load_v3(base), b = lm_load_v3(base + 20), c = lm_load_v3(base + 40);
    return lm_tri_test(a, b, c, q, tmin, tmax, t, u, v);
}
// The triangles of one leaf against a lane's ray, in leaf order: visit(slot, t, u, v) for every triangle hit inside (tmin, tmax); it returns true to stop (any-hit).
// `tested()` runs once per triangle tested (counting build).  LM_LEAF_PAIR (round 4, measured, off): bit 0 — the packets are fetched TWO at a time (six independent 16-byte
// loads in flight, one wait), so a leaf of c triangles costs ceil(c / 2) dependent round trips instead of c; bit 1 — the leaf's further 128-byte lines are touched at entry
// so that later iterations hit the L1.  Same tests on the same triangles in the same order: hit records cannot change (the GPU suite passes under every value).  All three
// are SLOWER on the frame (-1.7 %, -0.7 %, -4.2 %: profiles/r04_leaf_fetch_ab.txt): 12 more live registers spill in the 64-VGPR closest-hit kernel and cost the any-hit
// kernels a wave per SIMD, and a leaf's packets are contiguous, so most of its later fetches were L1 hits already.
#ifndef LM_LEAF_PAIR
#define LM_LEAF_PAIR 0
#endif
template <class Visit, class Tested>
__device__ __forceinline__ void lm_leaf_walk(const LmTriPacket* __restrict__ packets, uint32_t first, uint32_t count, const LmRayTri& rt, float tmin, float tmax,
                                             Visit visit, Tested tested)
{
#if LM_LEAF_PAIR & 2
    // touch the leaf's further 128-byte lines now (a leaf's packets are contiguous: <= 384 bytes), so that the later iterations' fetches hit the L1
    const char* base = (const char*)(packets + first);
    const uint32_t lastWord = count * 64u - 4u;
    uint32_t pf1 = *(const uint32_t*)(base + min(128u, lastWord)), pf2 = *(const uint32_t*)(base + min(256u, lastWord));
#endif
#if LM_LEAF_PAIR & 1
    for (uint32_t k = 0; k < count; k += 2u) {
        const uint32_t s0 = first + k, s1 = first + min(k + 1u, count - 1u);          // an odd leaf's last round fetches its last packet twice (same lines)
        const char* pa = (const char*)(packets + s0) + rt.row; const char* pb = (const char*)(packets + s1) + rt.row;
        LmV3 a0 = lm_load_v3(pa), a1 = lm_load_v3(pa + 20), a2 = lm_load_v3(pa + 40), b0 = lm_load_v3(pb), b1 = lm_load_v3(pb + 20), b2 = lm_load_v3(pb + 40);
        asm volatile("" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(b0.x), "+v"(b1.x), "+v"(b2.x));     // all six loads before the first use
        float t, u, v;
        tested();
        if (lm_tri_test(a0, a1, a2, rt, tmin, tmax, t, u, v) && visit(s0, t, u, v)) break;
        if (k + 1u < count) {
            tested();
            if (lm_tri_test(b0, b1, b2, rt, tmin, tmax, t, u, v) && visit(s1, t, u, v)) break;
        }
    }
#else
    for (uint32_t k = 0; k < count; k++) {
        float t, u, v;
        tested();
        if (lm_tri(packets, first + k, rt, tmin, tmax, t, u, v) && visit(first + k, t, u, v)) break;
    }
#endif
#if LM_LEAF_PAIR & 2
    asm volatile("" :: "v"(pf1), "v"(pf2));
#endif
}

__device__ __forceinline__ float lm_safe_rcp(float d)
{
    const float ooeps = 1e-20f;
    return 1.0f / (fabsf(d) > ooeps ? d : copysignf(ooeps, d));
}

// One step through a 4-wide node: slab-test the four quantised child boxes against [tmin, hitT], continue with the nearest
// hit child and push the others far-to-near (closest-hit) or in node order (any-hit).  Returns the next node / leaf
// reference, or LM_REF_NONE when the stack is empty.  `boxes` counts child boxes tested (instrumented build).
#ifndef LM_ANY_ORDERED
#define LM_ANY_ORDERED 0      // 1: any-hit queries also visit children near to far (finds close occluders sooner, costs the sort)
#endif
#ifndef LM_SLAB_PERM
#define LM_SLAB_PERM 3       // 0: min / max of the two plane distances per axis;
                             // 1: the near / far plane per axis picked by the sign of the direction (one v_perm_b32 on the packed lo|hi word), then 2 conversions
                             //    + 2 FMAs (measured alone against 0: closest hit 270 -> 257 us, visibility 244 -> 230, NEE shadow 98 -> 95; identical results);
                             // 3: the permute builds the FLOAT 2^23 + q directly (exponent byte of 2^23 over the 16-bit plane index: exact, no conversion), near
                             //    and far plane of an axis sit in a register pair and take ONE packed FMA: 3 instructions per axis instead of 5.  The offset
                             //    is folded into the ray's b (one rounding of b - 2^23 a: up to half a grid cell), which the builders' rounding margin
                             //    covers (LM_QUANT_MARGIN: bvh.cpp `quant`, kernels.hip lm_quant_axis)
#endif
// t = q * a + b per axis (dequantisation folded into the slab test); variant 3 keeps (a, b - 2^23 a) as a pair and the two byte selectors per axis
#ifndef LM_PERM_VGPR
#define LM_PERM_VGPR 0       // 1: the exponent word 0x4b000000 of the slab test's permutes sits in a VGPR instead of an SGPR.  profiles/r03_valu_peak.txt: a VOP3
#endif                       // instruction with an SGPR source issues at 4.2 cycles per SIMD, with VGPR / inline-constant sources at 2.4; six permutes per child box
struct LmRayQ { lm_f2 x, y, z; uint32_t nx, fx, ny, fy, nz, fz;
#if LM_PERM_VGPR
                uint32_t k23;
#endif
                uint32_t oct; };     // 8-wide tree: (direction octant) << 4 = byte offset of the slot a ray visits first; slot of visit p = p ^ octant
__device__ __forceinline__ void lm_ray_setup(const LmScene& sc, const lf3& o, const lf3& d, LmRayQ& r, LmRayTri& q)
{
    const float idx = lm_safe_rcp(d.x), idy = lm_safe_rcp(d.y), idz = lm_safe_rcp(d.z);
    // node boxes are 16-bit fixed point: world = qmin + q * qstep, so t = q * (qstep * idir) + (qmin - o) * idir
    const float ax = sc.quant[3] * idx, ay = sc.quant[4] * idy, az = sc.quant[5] * idz;
    const float bx = (sc.quant[0] - o.x) * idx, by = (sc.quant[1] - o.y) * idy, bz = (sc.quant[2] - o.z) * idz;
#if LM_SLAB_PERM == 3
    r.x = (lm_f2){ax, fmaf(-8388608.f, ax, bx)}; r.y = (lm_f2){ay, fmaf(-8388608.f, ay, by)}; r.z = (lm_f2){az, fmaf(-8388608.f, az, bz)};
    // v_perm_b32 selectors, result bytes 3..0: exponent byte of 2^23 (byte 7 of the source pair), 0x00, then the lo (a >= 0: near) or hi half of the word
    r.nx = ax < 0.f ? 0x070c0302u : 0x070c0100u; r.fx = r.nx ^ 0x00000202u;
    r.ny = ay < 0.f ? 0x070c0302u : 0x070c0100u; r.fy = r.ny ^ 0x00000202u;
    r.nz = az < 0.f ? 0x070c0302u : 0x070c0100u; r.fz = r.nz ^ 0x00000202u;
#else
    r.x = (lm_f2){ax, bx}; r.y = (lm_f2){ay, by}; r.z = (lm_f2){az, bz};
    // swapping the halves of the packed word by the sign gives (near, far) in (lo, hi)
    r.nx = ax < 0.f ? 0x01000302u : 0x03020100u; r.ny = ay < 0.f ? 0x01000302u : 0x03020100u; r.nz = az < 0.f ? 0x01000302u : 0x03020100u;
    r.fx = r.fy = r.fz = 0u;
#endif
#if LM_PERM_VGPR
    r.k23 = 0x4b000000u; asm volatile("" : "+v"(r.k23));           // opaque to the compiler: stays a VGPR
#endif
    r.oct = ((d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u)) << 4;
    // the triangle test's view of the ray (lm_tri_test): kz = dominant axis (ties: x before y before z), (kx, ky, kz) cyclic
    const float ax_ = fabsf(d.x), ay_ = fabsf(d.y), az_ = fabsf(d.z);
    const bool k0 = ax_ >= ay_ && ax_ >= az_, k1 = !k0 && ay_ >= az_;
    q.oxy = (lm_f2){k0 ? o.y : k1 ? o.z : o.x, k0 ? o.z : k1 ? o.x : o.y}; q.oz = k0 ? o.x : k1 ? o.y : o.z;
    q.sz = k0 ? idx : k1 ? idy : idz;
    q.sxy = (lm_f2){(k0 ? d.y : k1 ? d.z : d.x) * q.sz, (k0 ? d.z : k1 ? d.x : d.y) * q.sz};
    q.row = k0 ? 4u : k1 ? 8u : 0u;
}
// Slab test of one quantised child box against [tmin, hitT]: key = entry distance (its bit pattern orders like the value: tn >= tmin >= 0), or
// 0xffffffff for a miss.  An absent child carries an inverted box (lo = 0xffff, hi = 0: near > far on every axis), so it misses without a test.
__device__ __forceinline__ void lm_slab(const uint4& q, const LmRayQ& r, float tmin, float hitT, uint32_t& key)
{
#if LM_SLAB_PERM == 3
    // t is monotonic in q (fma rounds monotonically), rising for a >= 0 and falling for a < 0, so the smaller of the two plane distances is the
    // lo plane's for a >= 0 and the hi plane's otherwise
    lm_f2 px, py, pz, tx, ty, tz;
#if LM_PERM_VGPR
    const uint32_t e = r.k23;
#else
    const uint32_t e = 0x4b000000u;
#endif
    px.x = u2f(__builtin_amdgcn_perm(e, q.x, r.nx)); px.y = u2f(__builtin_amdgcn_perm(e, q.x, r.fx));
    py.x = u2f(__builtin_amdgcn_perm(e, q.y, r.ny)); py.y = u2f(__builtin_amdgcn_perm(e, q.y, r.fy));
    pz.x = u2f(__builtin_amdgcn_perm(e, q.z, r.nz)); pz.y = u2f(__builtin_amdgcn_perm(e, q.z, r.fz));
    // (near, far) * a + b': source 0 by halves, source 1 = the pair's low word for both results, source 2 = its high word for both
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tx) : "v"(px), "v"(r.x));
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(ty) : "v"(py), "v"(r.y));
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(tz) : "v"(pz), "v"(r.z));
    // (written as instructions: behind an asm result, fmaxf / fminf would first quiet a possible signalling NaN — 2 more instructions per child)
    float tn, tf;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(tx.x), "v"(ty.x), "v"(tz.x));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(tx.y), "v"(ty.y), "v"(tz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(tn) : "v"(tn), "v"(tmin));
    asm("v_min_f32 %0, %1, %2" : "=v"(tf) : "v"(tf), "v"(hitT));
#elif LM_SLAB_PERM == 1
    const uint32_t px = __builtin_amdgcn_perm(q.x, q.x, r.nx), py = __builtin_amdgcn_perm(q.y, q.y, r.ny), pz = __builtin_amdgcn_perm(q.z, q.z, r.nz);
    const float nx = fmaf((float)(px & 0xffffu), r.x.x, r.x.y), fx = fmaf((float)(px >> 16), r.x.x, r.x.y);
    const float ny = fmaf((float)(py & 0xffffu), r.y.x, r.y.y), fy = fmaf((float)(py >> 16), r.y.x, r.y.y);
    const float nz = fmaf((float)(pz & 0xffffu), r.z.x, r.z.y), fz = fmaf((float)(pz >> 16), r.z.x, r.z.y);
    const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
    const float tf = fminf(fminf(fx, fy), fminf(fz, hitT));
#else
    const float lox = fmaf((float)(q.x & 0xffffu), r.x.x, r.x.y), hix = fmaf((float)(q.x >> 16), r.x.x, r.x.y);
    const float loy = fmaf((float)(q.y & 0xffffu), r.y.x, r.y.y), hiy = fmaf((float)(q.y >> 16), r.y.x, r.y.y);
    const float loz = fmaf((float)(q.z & 0xffffu), r.z.x, r.z.y), hiz = fmaf((float)(q.z >> 16), r.z.x, r.z.y);
    const float tn = fmaxf(fmaxf(fminf(lox, hix), fminf(loy, hiy)), fmaxf(fminf(loz, hiz), tmin));
    const float tf = fminf(fminf(fmaxf(lox, hix), fmaxf(loy, hiy)), fminf(fmaxf(loz, hiz), hitT));
#endif
    key = tn <= tf ? f2u(tn) : 0xffffffffu;
}
__device__ __forceinline__ void lm_cex(uint32_t& ka, int& ra, uint32_t& kb, int& rb)
{
    const bool sw = kb < ka;
    const uint32_t k0 = min(ka, kb), k1 = max(ka, kb);
    const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}
#if LM_WIDTH == 8
#define LM_EXPERIMENTS_PART 1      // the 8-wide node step (round 3: measured, -5 ... -15 %)
#include "lm_experiments.h"
#else
template <bool ANY>
__device__ __forceinline__ int lm_node_step(const LmScene& sc, int cur, const LmRayQ& rq, float tmin, float hitT, const LmStack& stack, int& sp,
                                            const lm_lds_u4* top, uint32_t* boxes = nullptr)
{
    uint4 q0, q1, q2, q3;
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
    if (boxes) *boxes += ((int)q0.w != LM_REF_NONE) + ((int)q1.w != LM_REF_NONE) + ((int)q2.w != LM_REF_NONE) + ((int)q3.w != LM_REF_NONE);   // counting build
    uint32_t k0, k1, k2, k3;
    lm_slab(q0, rq, tmin, hitT, k0); lm_slab(q1, rq, tmin, hitT, k1); lm_slab(q2, rq, tmin, hitT, k2); lm_slab(q3, rq, tmin, hitT, k3);
    int r0 = (int)q0.w, r1 = (int)q1.w, r2 = (int)q2.w, r3 = (int)q3.w;
    if (!ANY || LM_ANY_ORDERED) {     // order the children by entry distance (a 5-comparator network; misses carry the largest key)
        lm_cex(k0, r0, k1, r1); lm_cex(k2, r2, k3, r3); lm_cex(k0, r0, k2, r2); lm_cex(k1, r1, k3, r3); lm_cex(k1, r1, k2, r2);
        if (k0 == 0xffffffffu) return sp == 0 ? LM_REF_NONE : lm_pop(stack, sp);
        if (k3 != 0xffffffffu) lm_push(stack, sp, r3);
        if (k2 != 0xffffffffu) lm_push(stack, sp, r2);
        if (k1 != 0xffffffffu) lm_push(stack, sp, r1);
        return r0;
    }
    int next = LM_REF_NONE;
    if (k3 != 0xffffffffu) next = r3;
    if (k2 != 0xffffffffu) { if (next != LM_REF_NONE) lm_push(stack, sp, next); next = r2; }
    if (k1 != 0xffffffffu) { if (next != LM_REF_NONE) lm_push(stack, sp, next); next = r1; }
    if (k0 != 0xffffffffu) { if (next != LM_REF_NONE) lm_push(stack, sp, next); next = r0; }
    if (next == LM_REF_NONE) return sp == 0 ? LM_REF_NONE : lm_pop(stack, sp);
    return next;
}

#ifndef LM_NODE_PIPELINE
#define LM_NODE_PIPELINE 0
#endif
#if LM_NODE_PIPELINE
#define LM_EXPERIMENTS_PART 2      // node fetch / evaluation in two halves (round 2: measured, -5 %)
#include "lm_experiments.h"
#endif
#endif   // LM_WIDTH

template <bool ANY>
__device__ __forceinline__ bool lm_traverse(const LmScene& sc, const lf3& o, const lf3& d, float tmin, float tmax,
                                            const LmStack& stack, LmHit& hit, uint32_t* cnt)
{
    LmRayQ rq;
    LmRayTri rt;
    lm_ray_setup(sc, o, d, rq, rt);
    float hitT = tmax;
    bool found = false;
    int sp = 0;
    int cur = 0;
#if LM_INSTRUMENT
    uint32_t nNodes = 0, nTris = 0;
#endif
    for (;;) {
        while (cur >= 0 && cur != LM_REF_NONE) {
#if LM_INSTRUMENT
            cur = lm_node_step<ANY>(sc, cur, rq, tmin, hitT, stack, sp, nullptr, &nNodes);      // + child boxes tested
#else
            cur = lm_node_step<ANY>(sc, cur, rq, tmin, hitT, stack, sp, nullptr);
#endif
        }
        if (cur == 0x7fffffff) break;
        // leaf
        const uint32_t leaf = (uint32_t)(~cur);
        const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        lm_leaf_walk(sc.packets, first, count, rt, tmin, tmax, [&](uint32_t slot, float t, float u, float v) {
            if (ANY) { found = true; return true; }
            if (t < hitT || (t == hitT && found && sc.triOrder[slot] < sc.triOrder[hit.slot])) {      // (the tie-break keys are fetched on a tie only)
                hitT = t; found = true;
                hit.t = t; hit.u = u; hit.v = v; hit.slot = slot;
            }
            return false;
        }, [&]() {
#if LM_INSTRUMENT
            nTris++;
#endif
        });
        if (ANY && found) break;
        if (sp == 0) break;
        cur = lm_pop(stack, sp);
    }
#if LM_INSTRUMENT
    atomicAdd((unsigned long long*)(cnt + LM_CNT_NODES), (unsigned long long)nNodes);
    atomicAdd((unsigned long long*)(cnt + LM_CNT_TRIS), (unsigned long long)nTris);
#endif
    return found;
}

// The same traversal with the query kind chosen PER LANE at run time (path tail in pair mode: half of a wavefront's lanes trace
// closest-hit continuation rays while the other half trace the any-hit shadow rays of the previous depth).  Any-hit lanes take the
// ordered node step too — an occlusion answer does not depend on the visiting order — and stop at their first hit.  A lane without a
// ray returns at once.  Results are those of lm_traverse<false> / lm_traverse<true>.
__device__ __forceinline__ bool lm_traverse_mixed(const LmScene& sc, const lf3& o, const lf3& d, float tmin, float tmax, bool any, bool valid,
                                                  const LmStack& stack, LmHit& hit, uint32_t* cnt)
{
    if (!valid) return false;
    LmRayQ rq;
    LmRayTri rt;
    lm_ray_setup(sc, o, d, rq, rt);
    float hitT = tmax;
    bool found = false;
    int sp = 0;
    int cur = 0;
#if LM_INSTRUMENT
    uint32_t nNodes = 0, nTris = 0;
#endif
    for (;;) {
        while (cur >= 0 && cur != LM_REF_NONE) {
#if LM_INSTRUMENT
            cur = lm_node_step<false>(sc, cur, rq, tmin, hitT, stack, sp, nullptr, &nNodes);
#else
            cur = lm_node_step<false>(sc, cur, rq, tmin, hitT, stack, sp, nullptr);
#endif
        }
        if (cur == 0x7fffffff) break;
        const uint32_t leaf = (uint32_t)(~cur);
        const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        lm_leaf_walk(sc.packets, first, count, rt, tmin, tmax, [&](uint32_t slot, float t, float u, float v) {
            if (any) { found = true; return true; }
            if (t < hitT || (t == hitT && found && sc.triOrder[slot] < sc.triOrder[hit.slot])) {      // (the tie-break keys are fetched on a tie only)
                hitT = t; found = true;
                hit.t = t; hit.u = u; hit.v = v; hit.slot = slot;
            }
            return false;
        }, [&]() {
#if LM_INSTRUMENT
            nTris++;
#endif
        });
        if (any && found) break;
        if (sp == 0) break;
        cur = lm_pop(stack, sp);
    }
#if LM_INSTRUMENT
    atomicAdd((unsigned long long*)(cnt + LM_CNT_NODES), (unsigned long long)nNodes);
    atomicAdd((unsigned long long*)(cnt + LM_CNT_TRIS), (unsigned long long)nTris);
#endif
    return found;
}

// ---------------------------------------------------------------------------------------------------------------------
// Queue traversal with per-lane ray replacement ("persistent threads").  Wavefront w of the launch owns the 64-ray groups
// w, w + W, w + 2W, ... of the queue (W = wavefronts in the grid; no atomics: one address retires only ~88 returning atomics
// per microsecond).  When fewer than `refillBelow` lanes of a wave are still traversing, the others take the next rays of
// the wave's groups, so incoherent rays of very different length do not leave most of the 64 lanes idle; refillBelow <= 1
// keeps a wave on one group at a time (best for coherent rays: an 8x8 pixel bundle stays together).
// `done(rayIndex, found, hit)` runs once per ray.  Results are identical to lm_traverse (same tests, same tie-break).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef LM_PRIO_RAYS
#define LM_PRIO_RAYS 1048576u
#endif
#ifndef LM_NODE_EXIT_FRAC
#define LM_NODE_EXIT_FRAC 3      // leave the node loop when fewer than a third of the round's lanes still descend (0: use the absolute LM_NODE_EXIT)
#endif
#ifndef LM_NODE_EXIT
#define LM_NODE_EXIT 14
#endif
#ifndef LM_SPECULATE
#define LM_SPECULATE 0       // 1: a lane that reaches its first leaf postpones it and keeps stepping through nodes (Aila & Laine's speculative
#endif                       //    while-while traversal) instead of idling until the wave leaves the node loop; hit records cannot change
template <bool ANY, class Fetch, class Done>
__device__ __forceinline__ void lm_trace_queue(const LmScene& sc, uint32_t n, int refillBelow, const LmStack& stack, const lm_lds_u4* top,
                                               uint32_t* cnt, Fetch fetch, Done done)
{
    const int root = (LM_TOP_NODES != 0 && top) ? LM_TOP_BASE : 0;
    // A small queue cannot fill the machine: its launch time is one wave's dependent chain, which stretches when the wave
    // shares its SIMD with VALU-bound kernels of the other streams.  Such waves ask the SIMD arbiter for priority.
    if (n < LM_PRIO_RAYS) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = lm_lane();
    const uint32_t W = gridDim.x * (LM_BLOCK / 64u);
    uint32_t group = blockIdx.x * (LM_BLOCK / 64u) + (threadIdx.x >> 6);      // wave-uniform
    uint32_t used = 0;                                                     // rays already handed out from `group` (wave-uniform)
    bool active = false;
    uint32_t rayIdx = 0;
    lf3 o = v3(0.f), d = v3(0.f);
    float tmin = 0.f, tmax = 0.f, hitT = 0.f;
    LmRayQ rq = {(lm_f2){0.f, 0.f}, (lm_f2){0.f, 0.f}, (lm_f2){0.f, 0.f}, 0u, 0u, 0u, 0u, 0u, 0u};
    LmRayTri rt = {(lm_f2){0.f, 0.f}, 0.f, (lm_f2){0.f, 0.f}, 0.f, 0u};
    bool found = false;
    int sp = 0, cur = 0;
#if LM_SPECULATE
    int pending = LM_REF_NONE;          // postponed leaf
#endif
    LmHit hit; hit.t = -1.f; hit.u = 0.f; hit.v = 0.f; hit.slot = 0;
#if LM_INSTRUMENT
    uint32_t nNodes = 0, nTris = 0, raySteps = 0;
#endif
    // triangles of one leaf against the lane's ray
    auto testLeaf = [&](int ref) {
        const uint32_t leaf = (uint32_t)(~ref);
        const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        lm_leaf_walk(sc.packets, first, count, rt, tmin, tmax, [&](uint32_t slot, float t, float u, float v) {
            if (ANY) { found = true; return true; }
            if (t < hitT || (t == hitT && found && sc.triOrder[slot] < sc.triOrder[hit.slot])) {      // (the tie-break keys are fetched on a tie only)
                hitT = t; found = true;
                hit.t = t; hit.u = u; hit.v = v; hit.slot = slot;
            }
            return false;
        }, [&]() {
#if LM_INSTRUMENT
            nTris++; raySteps++;
            { const unsigned long long m = __ballot(true);       // lane occupancy of this triangle-test issue
              if ((int)lane == __ffsll((long long)m) - 1) { atomicAdd((unsigned long long*)(cnt + LM_CNT_OCC + 4), (unsigned long long)__popcll(m)); atomicAdd((unsigned long long*)(cnt + LM_CNT_OCC + 6), 64ull); } }
#endif
        });
    };
    for (;;) {
        // ---- refill idle lanes from the wave's groups
        unsigned long long need = __ballot(!active);
        bool drained = (unsigned long long)group * 64ull >= (unsigned long long)n;
        while (need != 0ull && !drained) {
            const uint32_t base = group * 64u;
            const uint32_t avail = min(64u, n - base) - used;
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t give = min(want, avail);
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!active && rank < give) {
                rayIdx = base + used + rank;
                fetch(rayIdx, o, d, tmin, tmax);
                lm_ray_setup(sc, o, d, rq, rt);
                hitT = tmax; found = false; sp = 0; cur = root;
#if LM_SPECULATE
                pending = LM_REF_NONE;
#endif
                active = true;
#if LM_INSTRUMENT
                raySteps = 0;
#endif
            }
            used += give;
            if (used == min(64u, n - base)) { group += W; used = 0; drained = (unsigned long long)group * 64ull >= (unsigned long long)n; }
            need = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;
        // ---- traverse until the ray ends or the wave has become too empty
        while (active) {
#if LM_NODE_EXIT
            const int roundLanes = (int)__popcll(__ballot(true));
#endif
#if LM_NODE_PIPELINE && !LM_INSTRUMENT && LM_WIDTH == 4
            bool pre = false;                                        // nq0..nq3 hold the records of `cur` (only inside this loop: a lane that
            uint4 nq0, nq1, nq2, nq3;                                // leaves it with a prefetched node fetches it again on re-entry)
#endif
            while (cur >= 0 && cur != 0x7fffffff) {
#if LM_NODE_EXIT
                // leave the node loop once few lanes are still descending while others wait with a leaf (or a finished ray):
                // those test their triangles and rejoin, instead of idling until the slowest lane has found its leaf
#if LM_NODE_EXIT_FRAC
                { const int descending = (int)__popcll(__ballot(true)); if (descending * LM_NODE_EXIT_FRAC < roundLanes) break; }
#else
                { const int descending = (int)__popcll(__ballot(true)); if (descending < LM_NODE_EXIT && descending < roundLanes) break; }
#endif
#endif
#if LM_INSTRUMENT
                raySteps++;
                { const unsigned long long m = __ballot(true);           // lane occupancy of this node-step issue
                  if ((int)lane == __ffsll((long long)m) - 1) { atomicAdd((unsigned long long*)(cnt + LM_CNT_OCC), (unsigned long long)__popcll(m)); atomicAdd((unsigned long long*)(cnt + LM_CNT_OCC + 2), 64ull); } }
                cur = lm_node_step<ANY>(sc, cur, rq, tmin, hitT, stack, sp, top, &nNodes);      // + child boxes tested
#elif LM_NODE_PIPELINE && LM_WIDTH == 4
                if (!pre) lm_node_fetch(sc, cur, top, nq0, nq1, nq2, nq3);
                cur = lm_node_eval<ANY>(sc, nq0, nq1, nq2, nq3, pre, rq, tmin, hitT, stack, sp, top);
#else
                cur = lm_node_step<ANY>(sc, cur, rq, tmin, hitT, stack, sp, top);
#endif
#if LM_SPECULATE
                if (cur < 0 && pending == LM_REF_NONE) { pending = cur; cur = sp == 0 ? LM_REF_NONE : lm_pop(stack, sp); }
#endif
            }
#if LM_SPECULATE
            if (pending != LM_REF_NONE) {
                testLeaf(pending);
                pending = LM_REF_NONE;
                if (ANY && found) cur = LM_REF_NONE;
            }
#endif
            if (cur < 0) {
                testLeaf(cur);
                cur = ((ANY && found) || sp == 0) ? 0x7fffffff : lm_pop(stack, sp);
            }
            const bool fin = cur == 0x7fffffff;
            if (fin) {
                done(rayIdx, found, hit);
                active = false;
#if LM_INSTRUMENT
                // per-ray step histogram (4-wide nodes visited + triangles tested), log2 buckets, and the maximum
                atomicAdd(cnt + LM_CNT_STEP_HIST + min(15, 31 - __clz((int)(raySteps | 1u))), 1u);
                atomicMax(cnt + LM_CNT_STEP_MAX, raySteps);
#endif
            }
            // all lanes still in this loop vote (the ones that just finished included): when too few keep traversing,
            // they leave the loop with their state intact so that the idle lanes can take new rays
            const unsigned long long still = __ballot(!fin);
            if (!fin && !drained && (int)__popcll(still) < refillBelow) break;
        }
    }
#if LM_INSTRUMENT
    atomicAdd((unsigned long long*)(cnt + LM_CNT_NODES), (unsigned long long)nNodes);
    atomicAdd((unsigned long long*)(cnt + LM_CNT_TRIS), (unsigned long long)nTris);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Packet traversal for coherent bundles (primary rays: the 64 rays of a wavefront are an 8 x 8 pixel tile from one origin).
// The wavefront walks the tree as ONE: the current node is wave-uniform (fetched once, through the scalar / broadcast path), every
// lane slab-tests the node's four child boxes with its own ray, a child is visited if ANY lane hits it, the visiting order is that of
// the first lane that hits, and the stack is one shared LDS array per wavefront.  A lane whose ray misses a node's box misses the
// boxes below it too (children lie inside their parent), so no per-lane masks are kept.  Against the per-lane traversal this drops
// the per-lane order network, stack traffic and node addressing (85 instead of 180 VALU instructions per node visit) and pays with
// visiting the union of the lanes' nodes.  The hit rule is order independent (minimum t, then lowest global triangle index), so the
// hit records are those of lm_trace_queue bit for bit.
// ---------------------------------------------------------------------------------------------------------------------
#define LM_PACKET_STACK LM_STACK_DEPTH       // shared stack entries per wavefront (the builder bounds the tree by it)
__device__ __forceinline__ void lm_sort2u(uint32_t& ka, int& ra, uint32_t& kb, int& rb) { if (kb < ka) { const uint32_t k = ka; ka = kb; kb = k; const int r = ra; ra = rb; rb = r; } }
template <bool ANY, class Fetch, class Done>
__device__ __forceinline__ void lm_trace_packets(const LmScene& sc, uint32_t n, lm_lds_int* wstack, const lm_lds_u4* top, Fetch fetch, Done done)
{
    const uint32_t lane = lm_lane();
    const uint32_t W = gridDim.x * (LM_BLOCK / 64u);
    const int root = (LM_TOP_NODES != 0 && top) ? LM_TOP_BASE : 0;
    for (uint32_t group = blockIdx.x * (LM_BLOCK / 64u) + (threadIdx.x >> 6); (unsigned long long)group * 64ull < (unsigned long long)n; group += W) {
        const uint32_t i = group * 64u + lane;
        const bool valid = i < n;
        lf3 o = v3(0.f), d = v3(0.f, 0.f, 1.f);
        float tmin = 0.f, tmax = 0.f;
        if (valid) fetch(i, o, d, tmin, tmax);
        LmRayQ rq;
        LmRayTri rt;
        lm_ray_setup(sc, o, d, rq, rt);
        float hitT = valid ? tmax : -1.f;                          // a lane without a ray (or, any-hit, with its answer) fails every box test: tf < tmin
            bool found = false;
        LmHit hit; hit.t = -1.f; hit.u = 0.f; hit.v = 0.f; hit.slot = 0;
        int sp = 0, cur = root;                                    // wave-uniform
        for (;;) {
            cur = __builtin_amdgcn_readfirstlane(cur);
#if LM_WIDTH == 8
            if (cur >= 0) {
                // the wave visits the children in the octant order of its first ray (the rays of an 8 x 8 pixel tile almost always share
                // their octant; where they do not the order is merely less near-to-far for some lanes): no distances, no sorting
                const uint32_t uoct = (uint32_t)__builtin_amdgcn_readfirstlane((int)rq.oct) >> 4;
                uint4 q[8];
#if LM_TOP_NODES
                if (cur >= LM_TOP_BASE) {
                    const lm_lds_u4* nd = top + 8u * (uint32_t)(cur - LM_TOP_BASE);
#pragma unroll
                    for (uint32_t p = 0; p < 8u; p++) q[p] = lm_lds_read4(nd + (p ^ uoct));
                } else
#endif
                {
                    const uint4* nd = sc.nodes[cur].c;
#pragma unroll
                    for (uint32_t p = 0; p < 8u; p++) q[p] = nd[p ^ uoct];
                }
                uint32_t hits = 0u;                                   // bit p: some lane enters visit p's child (wave-uniform)
                int refs[8];
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    uint32_t k;
                    lm_slab(q[p], rq, tmin, hitT, k);
                    hits |= __ballot(k != 0xffffffffu) != 0ull ? 1u << p : 0u;
                    refs[p] = __builtin_amdgcn_readfirstlane((int)q[p].w);
                }
                if (hits == 0u) {
                    if (sp == 0) break;
                    cur = wstack[--sp];
                    continue;
                }
                const int first = __ffs((int)hits) - 1;
                if (lane == 0u) {                                  // far children first: the nearest is followed, the next nearest popped first
                    int s = sp;
#pragma unroll
                    for (int p = 7; p >= 1; p--) if (((hits >> p) & 1u) && p > first) wstack[s++] = refs[p];
                }
                sp += (int)__popc(hits) - 1;
                int nx = refs[7];
#pragma unroll
                for (int p = 6; p >= 0; p--) nx = ((hits >> p) & 1u) ? refs[p] : nx;
                cur = nx;
                continue;
            }
#else
            if (cur >= 0) {
                uint4 q0, q1, q2, q3;
#if LM_TOP_NODES
                if (cur >= LM_TOP_BASE) {
                    const lm_lds_u4* nd = top + 4u * (uint32_t)(cur - LM_TOP_BASE);
                    q0 = lm_lds_read4(nd); q1 = lm_lds_read4(nd + 1); q2 = lm_lds_read4(nd + 2); q3 = lm_lds_read4(nd + 3);
                } else
#endif
                {
                    const uint4* nd = sc.nodes[cur].c;
#ifndef LM_PACKET_SCALAR_NODE
#define LM_PACKET_SCALAR_NODE 1    // 1: the wave-uniform 64-byte node record is fetched ONCE through the scalar cache (s_load_dwordx16) instead of by four vector loads in which
#endif                             //    all 64 lanes ask for the same bytes (the compiler scalarises the leaf's triangle packets by itself, but not this load: `nodes` is written by the
                                   //    refit kernels, so it carries no read-only guarantee).  Scalar caches are invalidated at kernel boundaries: a refit in an earlier launch is seen.
                                   //    A/B: profiles/r04_packet_scalar_ab.txt
#if LM_PACKET_SCALAR_NODE && (defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__))      // the mnemonic and the single lgkmcnt counter are gfx9's: any other ARCH takes the vector loads
                    typedef uint32_t lm_u32x16 __attribute__((ext_vector_type(16)));
                    lm_u32x16 rec;
                    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(nd) : "memory");
                    q0 = make_uint4(rec[0], rec[1], rec[2], rec[3]); q1 = make_uint4(rec[4], rec[5], rec[6], rec[7]);
                    q2 = make_uint4(rec[8], rec[9], rec[10], rec[11]); q3 = make_uint4(rec[12], rec[13], rec[14], rec[15]);
#else
                    q0 = nd[0]; q1 = nd[1]; q2 = nd[2]; q3 = nd[3];
#endif
                }
                uint32_t k0, k1, k2, k3;
                lm_slab(q0, rq, tmin, hitT, k0); lm_slab(q1, rq, tmin, hitT, k1); lm_slab(q2, rq, tmin, hitT, k2); lm_slab(q3, rq, tmin, hitT, k3);
                // per child: does any lane enter it, and at what distance does the first such lane
                const unsigned long long m0 = __ballot(k0 != 0xffffffffu), m1 = __ballot(k1 != 0xffffffffu), m2 = __ballot(k2 != 0xffffffffu), m3 = __ballot(k3 != 0xffffffffu);
                uint32_t d0 = m0 ? (uint32_t)__builtin_amdgcn_readlane((int)k0, __ffsll((long long)m0) - 1) : 0xffffffffu;
                uint32_t d1 = m1 ? (uint32_t)__builtin_amdgcn_readlane((int)k1, __ffsll((long long)m1) - 1) : 0xffffffffu;
                uint32_t d2 = m2 ? (uint32_t)__builtin_amdgcn_readlane((int)k2, __ffsll((long long)m2) - 1) : 0xffffffffu;
                uint32_t d3 = m3 ? (uint32_t)__builtin_amdgcn_readlane((int)k3, __ffsll((long long)m3) - 1) : 0xffffffffu;
                int r0 = __builtin_amdgcn_readfirstlane((int)q0.w), r1 = __builtin_amdgcn_readfirstlane((int)q1.w), r2 = __builtin_amdgcn_readfirstlane((int)q2.w), r3 = __builtin_amdgcn_readfirstlane((int)q3.w);
                lm_sort2u(d0, r0, d1, r1); lm_sort2u(d2, r2, d3, r3); lm_sort2u(d0, r0, d2, r2); lm_sort2u(d1, r1, d3, r3); lm_sort2u(d1, r1, d2, r2);
                if (d0 == 0xffffffffu) {
                    if (sp == 0) break;
                    cur = wstack[--sp];
                    continue;
                }
                if (lane == 0u) {                                  // far children first: the nearest is followed, the next nearest popped first
                    int s = sp;
                    if (d3 != 0xffffffffu) wstack[s++] = r3;
                    if (d2 != 0xffffffffu) wstack[s++] = r2;
                    if (d1 != 0xffffffffu) wstack[s++] = r1;
                }
                sp += (d3 != 0xffffffffu) + (d2 != 0xffffffffu) + (d1 != 0xffffffffu);
                cur = r0;
                continue;
            }
#endif   // LM_WIDTH
            // leaf: every lane tests its ray against the leaf's triangles (a lane that cannot hit any more has hitT < tmin and fails the interval test)
            const uint32_t leaf = (uint32_t)(~cur);
            const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
            // the rays of a pixel tile almost always share their dominant axis, so the packet's three rows are selected wave-uniformly (scalar loads) and the
            // few wavefronts that straddle a diagonal direction take the leaf once per axis present
            unsigned long long todo = __ballot(true);
            while (todo != 0ull) {
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)rt.row, __ffsll((long long)todo) - 1);
                const bool mine = rt.row == row;
                todo &= ~__ballot(mine);
                for (uint32_t k = 0; k < count; k++) {
                    const char* base = (const char*)(sc.packets + first + k) + row;
                    const LmV3 a = lm_load_v3(base), b = lm_load_v3(base + 20), c = lm_load_v3(base + 40);
                    float t, u, v;
                    if (mine && lm_tri_test(a, b, c, rt, tmin, ANY ? hitT : tmax, t, u, v) && hitT >= tmin) {
                        if (ANY) { found = true; hitT = -1.f; continue; }
                        if (t < hitT || (t == hitT && found && sc.triOrder[first + k] < sc.triOrder[hit.slot])) {
                            hitT = t; found = true;
                            hit.t = t; hit.u = u; hit.v = v; hit.slot = first + k;
                        }
                    }
                }
            }
            if (ANY && __ballot(hitT >= tmin) == 0ull) break;       // every lane has its answer
            if (sp == 0) break;
            cur = wstack[--sp];
        }
        if (valid) done(i, found, hit);
    }
}
