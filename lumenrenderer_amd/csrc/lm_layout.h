// lm_layout.h — HBM data layout of the MI355X wavefront path tracer (shared by host C++ and HIP kernels).
//
// Everything the kernels stream is struct-of-arrays of 16-byte elements, so that a wavefront's 64 lanes read
// or write 1 KiB per instruction (DESIGN.md "Data layout").  The reference's AoS wire formats
// (IntersectionRayData 40 B, IntersectionData 16 B, SurfaceData 176 B, Reservoir 80 B, ShadowRayData 48 B —
// LumenPT/src/Shaders/CppCommon/WaveFrontDataStructs/*.h, ReSTIRData.h) carry the same fields.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LM_WIDTH
#define LM_WIDTH 4               // children per node of the tree the kernels traverse (4 or 8; the binary SAH tree is collapsed to this width, bvh.cpp).
#endif                           // 8: children sit in OCTANT slots (slot bit k set = the child lies on the + side of the node's centre along axis k) and a ray
                                 // visits slot (p ^ octant of its direction), p = 0 .. 7: near-to-far order without sorting, and half the dependent node
                                 // fetches of the 4-wide tree (which orders its children by entry distance with a comparator network)
#if LM_WIDTH != 4 && LM_WIDTH != 8
#error "LM_WIDTH must be 4 or 8"
#endif
#define LM_STACK_DEPTH (LM_WIDTH == 8 ? 96 : 64)   // traversal stack entries per lane; the BVH builder bounds the tree so that this suffices
#ifndef LM_STACK_LDS
#ifndef LM_STACK_LDS
#define LM_STACK_LDS 16          // of which in LDS; deeper entries spill to a per-thread global array
#endif
#endif
#define LM_BVH2_MAX_DEPTH 40     // depth bound of the binary tree the wide tree is collapsed from.  A wide node with n children pushes n - 1 entries and
                                 // spans at least ceil(log2 n) binary levels: (n - 1) / ceil(log2 n) <= 1.5 for n <= 4 and <= 7 / 3 for n <= 8, so the
                                 // stack need is <= 1.5 * 40 + 1 = 61 (4-wide) or 2.34 * 40 + 1 = 95 (8-wide) <= LM_STACK_DEPTH
#define LM_QUANT_MARGIN 2        // grid cells a quantised child box is widened by on each side beyond outward rounding: one for the fp32 evaluation of the slab
                                 // distances, one for the folded 2^23 offset of the packed slab test (lm_traverse.h LM_SLAB_PERM 3: up to half a cell)
#define LM_BOX_NONE 0x0000ffffu  // per-axis word of an absent child: lo = 0xffff, hi = 0 — inverted, so the slab test misses without looking at the reference
#define LM_REUSE_FLAGGED 0x80000000u
#define LM_REF_NONE 0x7fffffff   // absent child of a 4-wide node (also the traversal's "finished" marker; never followed)
#ifndef LM_TOP_NODES
#define LM_TOP_NODES (LM_WIDTH == 8 ? 9 : 21)   // top-of-tree node records (breadth-first from the root) the queue traversal kernels stage in LDS; 0 = none.
#endif                           // 21 four-wide records = three full levels = 1.3 KB per block beside the 16 KB stack (9 eight-wide records = two levels, 1.1 KB):
                                 // still eight blocks per CU
#define LM_TOP_BASE 0x40000000   // a node reference >= LM_TOP_BASE (and != LM_REF_NONE) names slot (ref - LM_TOP_BASE) of the staged table
#define LM_MAX_LEAF 8            // triangles per leaf representable in a leaf reference
#define LM_MAX_DEPTH 16          // path depth the counter block is sized for

// surface flags (reference: SurfaceData.h:18-24)
#define LM_SF_EMISSIVE 1u
#define LM_SF_ALPHA 2u
#define LM_SF_NON_INTERSECT 4u

// BVH2 node as the host builder produces it (64 bytes: the two children's fp32 boxes + references).  ref >= 0: inner node
// index; ref < 0: leaf, ~ref = (first_triangle << 3) | (count - 1).
struct LmNode {
    float4 n0;      // c0.lo.x c0.hi.x c0.lo.y c0.hi.y
    float4 n1;      // c1.lo.x c1.hi.x c1.lo.y c1.hi.y
    float4 n2;      // c0.lo.z c0.hi.z c1.lo.z c1.hi.z
    int4 ref;       // c0, c1, unused, unused
};
// Wide node as the GPU traverses it: LM_WIDTH children of 16 bytes (64 / 128 bytes: one cache line), the binary tree collapsed by surface
// area (bvh.cpp).  Child boxes are 16-bit fixed point relative to the scene box, rounded outward (lo | hi << 16 per axis); boxes only
// cull, so the hit record does not depend on them, nor on the width or the slot order.  An absent child has reference LM_REF_NONE and an
// inverted box.  8-wide: c[s] is octant slot s (see LM_WIDTH).
struct LmNodeW {
    uint4 c[LM_WIDTH];     // per child  x: lo.x | hi.x << 16   y: lo.y | hi.y << 16   z: lo.z | hi.z << 16   w: reference
};
// Triangle packet of the traversal, 64 bytes = one aligned cache line, in leaf order: the three world-space vertices as x y z x y each (lm_tri.h), so that
// a ray reads its cyclically permuted axes (kx, ky, kz: lm_traverse.h lm_tri_test) by one 12-byte load per vertex at float offset kx.  The all-zero packet
// (sentinel behind the last slot) never reports a hit: its three 2-D points coincide, every edge function and the determinant are zero for every ray.
struct LmTriPacket { float f[16]; };      // (64 bytes; device buffers come from hipMalloc: line-aligned.  No alignas: over-aligned host vectors went through memalign and fragmented the heap under topology edits — tools/soak.py)

// scene data table entry (reference: DevicePrimitiveInstance, ModelStructs.h:73-80)
struct LmEntry {
    float m[12];            // rows 0..2 of the row-major world matrix
    uint32_t vertBase;      // first vertex (in 48-byte vertices)
    uint32_t idxBase;       // first index
    uint32_t material;
    uint32_t mode;          // 0 ENABLED 1 DISABLED 2 OVERRIDE
    float4 emissive;        // override radiance rgb, w = scale
};
// device material (reference: DeviceMaterial, ModelStructs.h:33-63)
struct LmDevMaterial {
    float4 color, emissive, transmittance, tint;
    uint32_t p[4];
    int32_t tex[8];         // 0 clearcoat 1 clearcoatRoughness 2 transmission 3 diffuse 4 emissive 5 metalRoughness 6 normal 7 tint; -1 = null
    // Slots whose texture is a single texel (the default textures of a material without that map) or null are folded at material upload:
    // bit k of constMask set = texConst[k] IS what the fetch of slot k returns for every uv (same arithmetic, done once on the host), so
    // surface extraction skips descriptor + texel + sRGB-table loads for them.  Textures are immutable after lumen_mi_create_texture.
    uint32_t constMask, pad[3];
    float4 texConst[8];
};
struct LmTexDesc { uint32_t offset, w, h, srgb; };      // srgb: bit 0 = decode sRGB per texel, bit 1 = unquantised fp32 filter weights (lm_shade.h lm_tex2D)
// emissive triangle, 64 bytes, memory order of TriangleLight (LightData.h:21-27)
struct LmLight { float4 a, b, c, d; };   // a = p0.xyz p1.x | b = p1.yz p2.xy | c = p2.z n.xyz | d = radiance.xyz area

struct LmScene {
    LmNodeW* nodes;             // written only by the refit kernels
    const LmNodeW* top;         // LM_TOP_NODES records: the top of the tree in breadth-first order, child references relinked to table slots
                                // where the child is in the table too (lm_k_build_top, rebuilt whenever `nodes` changes)
    const float* quant;         // dequantisation of node boxes, in device memory so that a refit can move it without a host
                                // round trip: [0..2] qmin, [3..5] qstep (world = qmin + q * qstep), [6] box padding
    LmTriPacket* packets;               // written only by the refit kernels
    const uint2* triId;         // per BVH triangle slot: (table entry, primitive-local triangle), .x|0x80000000 never used
    const uint32_t* triOrder;   // per BVH triangle slot: global triangle index (tie-break key)
    const float4* verts;        // 3 float4 per vertex: (pos.xyz, uv.x) (uv.y, n.xyz) (tangent.xyzw)
    const uint32_t* indices;
    const LmEntry* entries;
    const LmDevMaterial* materials;
    const LmTexDesc* texDesc;
    const uint32_t* texels;     // RGBA8 pool
    const float* srgbLut;       // 256 entries
    const LmLight* lights;      // sorted by mean radiance
    const float* cdf;
    int* spill;                 // per-thread stack overflow area: (LM_STACK_DEPTH - LM_STACK_LDS) ints per thread of the largest trace grid
    uint32_t numLights;
    float cdfSum;
    uint32_t numEntries, numMaterials;   // sizes of `entries` / `materials` (small tables are staged in LDS by the extraction kernels)
};

// per-frame working set; all per-pixel arrays are indexed by the window-local pixel index
struct LmFrame {
    uint32_t W, H;              // full image
    uint32_t x0, y0, ww, wh;    // render window (tile + halo) inside the image
    uint32_t tx0, ty0, tx1, ty1; // the tile this renderer owns, window-local [tx0,tx1) x [ty0,ty1) (= the window when not tiled): work whose
                                // result is only needed for owned pixels (indirect waves, second reuse pass, ...) is skipped in the halo
    uint32_t n;                 // ww * wh
    // ray queues (ping-pong): origin.xyz | dir.xyz + local pixel index | contribution.xyz
    float4 *rayO[2], *rayD[2], *rayC[2];
    uint4* hits;                // entry, prim, half2 barycentrics, t bits
    // shadow-ray queue of the current wave: origin.xyz + tmax | dir.xyz + pixel | radiance.xyz
    float4 *shO, *shD, *shR;
    float4 *visO, *visD;        // ReSTIR visibility-ray queue, pass 1 (own buffers: runs concurrently with the NEE shadow queue)
    float4 *vis2O, *vis2D;      // pass 2 (filled by the temporal kernel, traced beside the second spatial pass)
    // depth-0 surface data: one 128-byte record (8 float4) per pixel + a 16-byte reuse-probe plane (kernels.hip).  Three sets
    // (current, previous, and the one the NEXT frame's extraction already fills while this frame's temporal pass still reads)
    float4* gbuf[3];
    float4* probe[3];
    uint32_t* rareTile[3];      // fast ReSTIR mode, per G-buffer set: one word per 16 x 16 pixel tile of the WINDOW (row-major, (ww + 15) / 16 per row), non-zero when the tile holds a depth-0
                                // surface outside the contracted evaluation (written by the extraction, cleared at the frame's start): the second, exact launch of every ReSTIR pass leaves
                                // blocks whose neighbourhood holds none at once instead of loading every pixel's surface to find that out.  NULL = not tracked (every block looks)
    uint32_t* reuseMask;        // per pixel: which of its five spatial-reuse candidates passed the similarity test (bits 0..4), or LM_REUSE_FLAGGED; written
                                // by the first spatial pass, read by the second (both draw the same candidates: the reference passes one seed to both)
    int fuseRc; uint32_t fuseSeed;      // read by the fused second spatial pass only (lm_k_restir_spatial*_fused): the reservoir buffer and the seed of the combine it ends with
    // reservoirs, 5 buffers (the reference's two swap-chain and two spatial buffers + [4], fresh candidates when candidate
    // generation runs ahead on its own stream): one 64-byte hot record (4 float4) per pixel + a contribution plane
    float4* res[5];
    float4* resC[5];
    uint32_t* motion;           // half2 motion vector per pixel (of this frame; double-buffered by frame parity on the host side)
    float4 *direct, *indirect;  // fp32 light channels
    float4* combined;           // merged / blended radiance
    uchar4* output;             // sRGB8
    uint32_t* counters;         // see LM_CNT_*
    unsigned long long* totals; // LM_CNT_WORDS + 1 running sums: the merge kernel adds every frame's counter block (and 1 to the last word), so that a
                                // throughput measurement counts the rays of ALL the frames it timed without reading counters back between frames
    int* swap;                  // ReSTIR swap-chain index (ReSTIR::m_SwapChainIndex), on the device: it advances once per EXECUTED
                                // wave, and the wave loop ends when a wave's queue is empty (WaveFrontRenderer.cpp:697,827) — a count only
                                // the device knows without a host round trip.  Kernels take buffer indices as LM_RES_* codes.
                                // swap[1] = waves the last frame executed (multi-GPU: ranks agree on the maximum, lm_k_wave_sync).
                                // swap[2], swap[3] = has swap-chain buffer 0 / 1 been written since the reservoirs were reset?  With an even number of
                                // executed waves per frame the "previous" buffer of the temporal pass never is: the pass then takes the
                                // reset reservoir it would load (all zero) without loading it.
                                // swap[4..9]: deferred history passes (frame.cpp "lazy reuse"): [4] = swap-chain index that was the FRONT buffer of the last
                                // merged frame, [5] = 1 when that frame's spatial / combine passes have run or are settled (not owed), [7] = that frame's
                                // LM_CNT_RARE flag, [8] = deferred executions, [9] = entries completed by lm_k_reuse_counts since the reservoirs were reset
    uint32_t* hazardList;       // lazy reuse: pixels (window-local) that were reuse surfaces in the previous frame (G-buffer set owedSet, -1 = nothing pending) and are
    int owedSet;                // flagged in this one, appended by the extraction; LM_CNT_HAZARD counts them (kernels.hip lm_k_reuse_counts)
    int deferred;               // history-building passes (both spatial passes, combine): 0 = launched with their frame; 1 / 2 = launched later, inside the
                                // next frame / between frames, and run only if the swap chain has turned (kernels.hip lm_reuse_owed)
    uint2* bags;                // 50 x 1000 light-bag entries: (light index, pdf bits)
};
// reservoir buffer index codes of the ReSTIR kernels: a literal index >= 0, or the swap-chain front / back buffer
#define LM_RES_CUR (-1)
#define LM_RES_PREV (-2)
#define LM_RES_OWED (-3)                         // the front buffer of the last merged frame (swap[4]): what a deferred history pass completes
// counter block layout (uint32 each)
#define LM_CNT_RAYS(d) (d)                       // rays entering wave d            [0, LM_MAX_DEPTH]
#define LM_CNT_SHADOW(d) (32 + (d))              // NEE shadow rays emitted by wave d
#define LM_CNT_RESTIR(p) (70 + (p))              // ReSTIR visibility rays of pass p (0, 1)
#define LM_CNT_RARE 72                           // flag: 1 when a depth-0 surface of the frame has a lobe outside the contracted evaluation (lm_bsdf.h lm_quick_contracts)
#define LM_CNT_HAZARD 73                         // lazy reuse: entries of LmFrame::hazardList
#define LM_CNT_STEP_HIST 96                       // instrumented build only: 16 log2 buckets of per-ray traversal steps (queue kernels)
#define LM_CNT_STEP_MAX 112                       // instrumented build only: longest per-ray traversal (steps)
#define LM_CNT_NODES 66                          // instrumented build only: child boxes slab-tested (u64 as 2 words); 2 boxes = one binary node of SURVEY 8 d4
#define LM_CNT_TRIS 68                           // instrumented build only: triangles tested (u64 as 2 words)
#define LM_CNT_OCC 120                            // instrumented build only, u64 each: active lanes / lane slots of node steps, of triangle tests
#define LM_CNT_WORDS 136
#ifndef LM_PRIMARY_CLEARS
#define LM_PRIMARY_CLEARS 1     // the frame's first kernel (lm_k_primary, block 0) zeroes the counter block: no fill launch in front of every frame on the wave stream
#endif

struct LmCamera { float eye[3], U[3], V[3], Wv[3]; float prevViewProj[16]; };
