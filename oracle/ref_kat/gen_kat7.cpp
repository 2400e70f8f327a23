// gen_kat7.cpp — known-answer generator, seventh translation unit: the two callees of the hot path that store radiance as binary16 —
// ShadeReservoirs (ReSTIRKernels.cu:619-665: a reservoir's contribution added to the half4 DIRECT surface) and MergeOutputChannels
// (WaveFrontKernels/GPUMergeOutputChannels.cu:5-88: channel sum, volumetric blend, running-mean blend, all in half4) — compiled from the reference's OWN TEXT and run
// thread by thread on the host (container-only; this file contains no reference source text).  make_kat.py slices those two line ranges (and ReSTIRKernels.cuh:17-18,
// WaveFrontDataStructs.h:13: the index macros) into /tmp/lumen_k7_*.inc, never into the repository; the reference's Half4.h is included as it lies.
//
// What this file supplies is what nvcc supplies to those lines:
//   * blockIdx / blockDim / threadIdx (plain variables, looped over: one call of the kernel body per thread);
//   * surf2Dread / surf2Dwrite<ushort4> on host arrays (a surface handle is an index into a small table);
//   * the five binary16 intrinsics Half4.h calls, which the vendored cuda_fp16.h declares for device code only: __hadd2, __hmul2, __h2div, __ushort_as_half,
//     __half_as_ushort.  Each is defined below as THE OPERATION ITS DOCUMENTATION STATES — the IEEE binary16 sum / product / quotient, round to nearest even — computed
//     as the binary32 operation on the header's own host __half2float values, rounded once by the header's own host __float2half.  That is exact, not an approximation:
//     rounding to p = 24 bits and then to p = 11 bits equals rounding once to 11 bits for +, x, / because 24 >= 2 * 11 + 2 (the double-rounding theorem).  [What is NOT
//     modelled: the hardware's half-division sequence (rcp.approx + one fix-up, cuda_fp16.hpp __hdiv) is documented as round-to-nearest but is not proven correctly
//     rounded for every operand pair; the divisor here is always a small integer frame count.]
// D1 replaces this fp16 arithmetic by fp32 in the product and the oracle; these rows let the tests say by HOW MUCH the two differ per operation instead of
// pricing the reference side with a formula (VERDICT r5 missing #4).
//
// Rows (32-bit words; halves as their 16-bit patterns, floats as bit patterns):
//   shd7  i | colour in (4 halves) weight contribution(3) | colour out (4 halves)
//   mrg7  i blend blendCount | DIRECT(4) INDIRECT(4) SPECULAR(4) VOLUMETRIC(4) old output(4) | new output(4)
#include <cmath>
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
using std::min; using std::max;
#include <cuda_runtime.h>
#include <sutil/vec_math.h>
#include <cuda_fp16.h>

// ---- what nvcc supplies ------------------------------------------------------------------------------------------------------------------
static uint3 blockIdx, threadIdx;
static dim3 blockDim, gridDim;
static inline __half kat_h(float f) { return __float2half(f); }
static inline __half2 __hadd2(const __half2 a, const __half2 b) { __half2 r; r.x = kat_h(__half2float(a.x) + __half2float(b.x)); r.y = kat_h(__half2float(a.y) + __half2float(b.y)); return r; }
static inline __half2 __hmul2(const __half2 a, const __half2 b) { __half2 r; r.x = kat_h(__half2float(a.x) * __half2float(b.x)); r.y = kat_h(__half2float(a.y) * __half2float(b.y)); return r; }
static inline __half2 __h2div(const __half2 a, const __half2 b) { __half2 r; r.x = kat_h(__half2float(a.x) / __half2float(b.x)); r.y = kat_h(__half2float(a.y) / __half2float(b.y)); return r; }
static inline __half __ushort_as_half(const unsigned short i) { __half h; memcpy(&h, &i, 2); return h; }
static inline unsigned short __half_as_ushort(const __half h) { unsigned short i; memcpy(&i, &h, 2); return i; }
struct KatSurface { ushort4* px; unsigned w, h; };
static std::vector<KatSurface> g_surfaces;                               // handle = index + 1
template <class T> static inline void surf2Dread(T* out, cudaSurfaceObject_t s, int xBytes, int y, int /*cudaBoundaryModeTrap*/)
{
    static_assert(sizeof(T) == sizeof(ushort4), "half4 surfaces only");
    const KatSurface& q = g_surfaces[(size_t)s - 1];
    memcpy(out, &q.px[(size_t)y * q.w + (size_t)xBytes / sizeof(T)], sizeof(T));
}
template <class T> static inline void surf2Dwrite(T v, cudaSurfaceObject_t s, int xBytes, int y, int /*cudaBoundaryModeTrap*/)
{
    static_assert(sizeof(T) == sizeof(ushort4), "half4 surfaces only");
    const KatSurface& q = g_surfaces[(size_t)s - 1];
    memcpy(&q.px[(size_t)y * q.w + (size_t)xBytes / sizeof(T)], &v, sizeof(T));
}

#include "Shaders/CppCommon/ReSTIRData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/LightData.h"
#include "Shaders/CppCommon/ArrayParameter.h"
#include "Shaders/CppCommon/CudaDefines.h"
#define __CUDACC__ 1                     // Half4.h defines its operators for device code only
#include "Shaders/CppCommon/Half4.h"
using namespace WaveFront;
#include "/tmp/lumen_k7_macros.inc"
#include "/tmp/lumen_k7_pdi.inc"
#undef CPU_ON_GPU
#define CPU_ON_GPU static                // __global__ has no meaning on the host: the kernel is an ordinary function called once per thread
#define device_launch_parameters_h_kat 1
#include "/tmp/lumen_k7_shade.inc"       // ShadeReservoirs
#include "/tmp/lumen_k7_merge.inc"       // MergeOutputChannels

static std::mt19937 rng(20261006u);
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(rng); }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static ushort4 randomHalf4(float scale, float zeroShare)
{
    auto one = [&]() { const float r = U(); return r < zeroShare ? 0.f : scale * r * r * (U() < 0.05f ? 40.f : 1.f); };
    half4 h(one(), one(), one(), U() < 0.5f ? 0.f : 1.f);
    return h.AsUshort4();
}

int main()
{
    const unsigned W = 64, H = 48, N = W * H;
    // ---- ShadeReservoirs: one reservoir per pixel (numReservoirsPerPixel = 1), output surface pre-filled
    {
        std::vector<ushort4> out(N);
        std::vector<Reservoir> res(N);
        g_surfaces.clear(); g_surfaces.push_back({out.data(), W, H});
        for (unsigned i = 0; i < N; i++) {
            out[i] = randomHalf4(2.0f, 0.3f);
            res[i] = Reservoir();
            const float r = U();
            res[i].weight = r < 0.15f ? 0.f : r < 0.2f ? -1.f : (r < 0.3f ? 60.f : 3.f) * U();
            res[i].sample.unshadowedPathContribution = make_float3(4.f * U() * U(), 4.f * U() * U(), 4.f * U() * U());
            if (U() < 0.05f) res[i].sample.unshadowedPathContribution = make_float3(3000.f * U(), 3000.f * U(), 3000.f * U());      // towards the binary16 range limit
        }
        const std::vector<ushort4> before = out;
        for (unsigned y = 0; y < H; y++) for (unsigned x = 0; x < W; x++) ShadeReservoirs(res.data(), W, x, y, x, y, 1);
        for (unsigned i = 0; i < N; i++) {
            const float3 c = res[i].sample.unshadowedPathContribution;
            printf("shd7 %u %u %u %u %u %u %u %u %u %u %u %u %u\n", i, before[i].x, before[i].y, before[i].z, before[i].w, bits(res[i].weight), bits(c.x), bits(c.y), bits(c.z),
                   out[i].x, out[i].y, out[i].z, out[i].w);
        }
    }
    // ---- MergeOutputChannels: four channel surfaces + the output surface, with and without blending, blend counts 0..9
    for (int pass = 0; pass < 3; pass++) {
        std::vector<ushort4> ch[4], out(N);
        g_surfaces.clear();
        for (int c = 0; c < 4; c++) { ch[c].resize(N); g_surfaces.push_back({ch[c].data(), W, H}); }
        g_surfaces.push_back({out.data(), W, H});
        const bool blend = pass > 0;
        const unsigned blendCount = pass == 0 ? 0u : pass == 1 ? 3u : 9u;
        for (unsigned i = 0; i < N; i++) {
            for (int c = 0; c < 4; c++) ch[c][i] = randomHalf4(c == 0 ? 3.f : 1.f, 0.2f);
            ch[2][i] = half4(0.f).AsUshort4();                                                          // SPECULAR: never written by the wavefront path (zero)
            ch[3][i] = half4(0.f).AsUshort4();                                                          // VOLUMETRIC: out of scope (no volume: alpha 0)
            out[i] = randomHalf4(2.0f, 0.1f);
        }
        const std::vector<ushort4> before = out;
        ArrayParameter<cudaSurfaceObject_t, static_cast<unsigned>(LightChannel::NUM_CHANNELS)> in;
        for (int c = 0; c < 4; c++) in[c] = (cudaSurfaceObject_t)(c + 1);
        gridDim = dim3((W + 15) / 16, (H + 15) / 16, 1); blockDim = dim3(16, 16, 1);
        for (unsigned j = 0; j < gridDim.y; j++) for (unsigned i = 0; i < gridDim.x; i++) for (unsigned v = 0; v < 16; v++) for (unsigned u = 0; u < 16; u++) {
            blockIdx = make_uint3(i, j, 0); threadIdx = make_uint3(u, v, 0);
            MergeOutputChannels(make_uint2(W, H), in, (cudaSurfaceObject_t)5, blend, blendCount);
        }
        for (unsigned i = 0; i < N; i++) {
            printf("mrg7 %u %d %u", i, blend ? 1 : 0, blendCount);
            for (int c = 0; c < 4; c++) printf(" %u %u %u %u", ch[c][i].x, ch[c][i].y, ch[c][i].z, ch[c][i].w);
            printf(" %u %u %u %u %u %u %u %u\n", before[i].x, before[i].y, before[i].z, before[i].w, out[i].x, out[i].y, out[i].z, out[i].w);
        }
    }
    return 0;
}
