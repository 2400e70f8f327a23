// lm_launch.h — host-callable launch table of kernels.hip (built twice: plain and node/triangle-counting).
#pragma once
#include "lm_layout.h"

struct LmKernelTable {
    void (*primary)(hipStream_t, int grid, LmFrame, LmCamera, uint32_t frameCount);
    void (*trace_closest)(hipStream_t, int grid, LmScene, const float4* o, const float4* d, const uint32_t* count, uint4* hits, float tmin, float tmax, uint32_t* counters, int refillBelow,
                          const float* eye /* with o == NULL: the common origin of the rays */);
    void (*extract0)(hipStream_t, int grid, LmScene, LmFrame, LmCamera, int cur, uint32_t seed2, int doIndirect, int outQ, uint32_t* outCount);
    void (*shade_wave)(hipStream_t, int grid, LmScene, LmFrame, int inQ, const uint32_t* inCount, uint32_t seed, uint32_t seed2, int doIndirect, uint32_t* outCount, uint32_t* shadowCount);
    void (*trace_shadow)(hipStream_t, int grid, LmScene, LmFrame, const uint32_t* count, float tmin, int refillBelow);
    void (*path_tail)(hipStream_t, int grid, LmScene, LmFrame, int inQ, const uint32_t* inCount, int depth0, int depthMax, uint32_t seed0, int lanesPerWave);
    void (*fill_bags)(hipStream_t, LmScene, LmFrame, uint32_t seed, uint32_t total);
    void (*pick_primary)(hipStream_t, int tiles, LmScene, LmFrame, int cur, int rc, uint32_t seed, uint32_t* visCount, int fast);
    void (*trace_shade)(hipStream_t, int grid, LmScene, LmFrame, int rc, const uint32_t* count, int refillBelow, int pass);
    void (*temporal)(hipStream_t, int tiles, LmFrame, int cur, int prev, int rc, int rp, int rf, uint32_t seed, uint32_t* visCount, int fast);
    void (*spatial)(hipStream_t, int tiles, LmFrame, int cur, int rin, int rout, uint32_t seed, int margin, int pass /* 0 first, 1 second */, int fast);
    void (*combine)(hipStream_t, int tiles, LmFrame, int cur, int rc, int rs, uint32_t seed, int fast);     // fast: 0 exact; 1 contracted evaluation with hardware rcp / rsq / sqrt (LmFast / LmQuick, lm_bsdf.h);
                                                                                                 // 2 the same + the second, exact launch for surfaces the contracted evaluation does not cover
    void (*clear)(hipStream_t, int grid, float4* p, uint32_t n);
    void (*merge)(hipStream_t, int grid, LmFrame, int blend, uint32_t blendCount, int depthMax);
    void (*query_any)(hipStream_t, int grid, LmScene, const float4* o, const float4* d, uint32_t n, float tmin, uint32_t* occluded, uint32_t* counters);
    void (*query_closest)(hipStream_t, int grid, LmScene, const float4* o, const float4* d, uint32_t n, float tmin, float tmax, uint4* id, float4* uvt, uint32_t* counters);
    void (*export_aux)(hipStream_t, int grid, LmFrame, int cur, float minD, float maxD, float* depth, uint2* normalRoughness);
    void (*refit_tris)(hipStream_t, LmScene, uint32_t nSlots, float4* triBox, uint32_t* bounds);
    void (*refit_quant)(hipStream_t, uint32_t* bounds, float* quant);
    void (*refit_level)(hipStream_t, LmScene, const uint32_t* levelNodes, uint32_t count, const float4* triBox, float4* nodeBox);
    void (*test_bsdf)(hipStream_t, uint32_t n, int mode, const float* mat, const float* N, const float* T, const float* wo, const float* aux, float* out);
    void (*test_math)(hipStream_t, uint32_t n, int fn, const float* x, const float* y, float* out);
    void (*spin)(hipStream_t, uint32_t ticks);        // one idle wavefront for `ticks` of the 100 MHz wall clock (schedule fuzzing, frame.cpp)
    void (*history_copy)(hipStream_t, int grid, LmFrame, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, float4* buf, int import);
    void (*wave_sync)(hipStream_t, int* swap, int* io, int import);
    void (*test_restir)(hipStream_t, int mode, uint32_t n, const float* a, const float* b, const uint32_t* c, uint32_t m, float* out);
    void (*build_top)(hipStream_t, const LmNodeW* nodes, LmNodeW* top);       // top-of-tree table of the queue traversal kernels (after every change of `nodes`)
    void (*export_half4)(hipStream_t, int grid, const float4* src, uint2* dst, uint32_t n);      // merged radiance rounded to the reference's half4 storage
    // counting sort of a ray queue by (origin cell, direction octant) into another queue; bins: 2 x 4096 words, zero on entry and on return
    void (*sort_rays)(hipStream_t, int grid, LmScene, const float4* srcO, const float4* srcD, const float4* srcC, float4* dstO, float4* dstD, float4* dstC,
                      const uint32_t* count, uint32_t* bins);
    void (*reuse_settle)(hipStream_t, LmFrame);       // after deferred history passes launched BETWEEN frames (kernels.hip lm_reuse_owed)
    void (*reuse_counts)(hipStream_t, LmFrame previous, int was, const uint32_t* list, const uint32_t* listCount, uint32_t seed);      // lazy reuse: completes the entries that outlive a dropped history pass
    void (*trace_primary)(hipStream_t, int grid, LmScene, LmFrame, LmCamera, uint32_t frameCount, uint4* hits, float tmin, float tmax);   // primary rays generated inside the packet traversal
    // known-answer hooks for whole kernels (kat.cpp; rows of tests/golden/ref_kat5.npz): synthetic surfaces / reservoirs laid out by the product's own store functions,
    // a visibility queue resolved from a mask, ShadeDirect / ShadeIndirect on surface rows
    void (*kat_pack_surfaces)(hipStream_t, const uint32_t* rows40, uint32_t n, float4* gbuf, float4* probe);
    void (*kat_reservoirs)(hipStream_t, uint32_t* rows17, uint32_t n, float4* hot, float4* contrib, int unpack);
    void (*kat_resolve)(hipStream_t, LmFrame, int rc, const uint32_t* count, const uint8_t* occluded, int pass);
    void (*kat_shade)(hipStream_t, LmScene, uint32_t n, uint32_t W, const uint32_t* rows43, int fast, uint32_t* direct12, uint32_t* indirect10);
    void (*kat_extract)(hipStream_t, LmScene, uint32_t n, const uint32_t* hits9, const uint32_t* rays9, uint32_t* out35);      // lm_extract on (hit, ray) rows against the current scene
    void (*kat_tex2d)(hipStream_t, LmScene, uint32_t n, int id, const float2* uv, float4* out);                                // lm_tex2D on one texture of the current scene
    void (*copy_rect)(hipStream_t, int grid, float4* dst, uint32_t dstPitch, const float4* src, uint32_t srcPitch, uint32_t w, uint32_t h);      // pitched RGBA32F rectangle (tile gather)
};
extern "C" const LmKernelTable* lm_kernel_table();
extern "C" const LmKernelTable* lm_kernel_table_noslp();        // the same kernels compiled with -fno-slp-vectorize (kernels.hip LM_NOSLP_VARIANT)
extern "C" const LmKernelTable* lm_kernel_table_instrumented();
