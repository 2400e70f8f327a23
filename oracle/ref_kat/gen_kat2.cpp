// Golden-vector generator, part 2: the remaining host-compilable pieces of the reference's hot path —
//   Reservoir::Update / UpdateWeight / Reset and CDF::Insert / Get / BinarySearch   (Shaders/CppCommon/ReSTIRData.h:115-178, 181-306)
//   make_color / toSRGB / quantizeUnsigned8Bits                                     (vendor/Include/Cuda/cuda/helpers.h:35-66)
//   __float2half / __half2float as the vendored CUDA headers define them on the host (what half4(...) / AsFloat4() of Half4.h:9-96 call)
// ReSTIRData.h must be included while __CUDACC__ is undefined (it pulls optix.h), hence a second translation unit beside gen_kat.cpp.
// half4's ARITHMETIC (Half4.h:100-206: __hadd2 / __hmul2 / __h2div) and everything in RenderingUtility.h are not pinned: the former
// are device-only intrinsics in the vendored cuda_fp16.h (:2126-2186, no host definition), the latter is dead code on this path
// (its only call sites are commented out: GPUShadeDirect.cu:128, GPUShadeIndirect.cu:55,86, ReSTIRKernels.cu:1288).
// Container-only; contains no reference source text — it only #includes it from /root/reference.
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <random>
using std::min; using std::max; using std::abs;
#include <cuda_runtime.h>
#include <sutil/vec_math.h>
#include "Shaders/CppCommon/ReSTIRData.h"
#include <cuda_fp16.h>

static std::mt19937 rng(20261003u);
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(rng); }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main()
{
    // ---- reservoir rows: 8 updates each.  in: per update (weight, solidAnglePdf, seed); out: per update (weightSum bits, sampleCount,
    // id of the held sample, returned bool), then weight bits after UpdateWeight, then state after Reset
    for (int row = 0; row < 400; row++) {
        Reservoir r;
        printf("resv");
        float w[8], p[8]; uint32_t sd[8];
        for (int k = 0; k < 8; k++) {
            const int kind = (row + k) % 7;
            w[k] = kind == 0 ? 0.f : kind == 1 ? U() * 1e-6f : kind == 2 ? U() * 1e4f : U();      // zero weights, tiny, huge, ordinary
            p[k] = kind == 3 ? 0.f : U();                                                        // pdf 0 exercises the MINFLOAT clamp
            sd[k] = (row % 5 == 0) ? 12345u + (uint32_t)row : rng();                              // same seed for every update (by-value quirk) or not
            printf(" %u %u %u", bits(w[k]), bits(p[k]), sd[k]);
        }
        for (int k = 0; k < 8; k++) {
            LightSample s; s.area = (float)(k + 1); s.solidAnglePdf = p[k];
            const bool took = r.Update(s, w[k], sd[k]);
            printf(" %u %lld %d %d", bits(r.weightSum), r.sampleCount, (int)r.sample.area, took ? 1 : 0);
        }
        r.UpdateWeight();
        printf(" %u", bits(r.weight));
        r.Reset();
        printf(" %u %lld %u %d\n", bits(r.weightSum), r.sampleCount, bits(r.weight), (int)r.sample.area);
    }
    // ---- CDF rows.  cdfw: id n data[0..63] (as CDF::Insert accumulates them, float running sum); cdfq: id value index pdf-bits
    const int sizes[] = {1, 2, 3, 5, 8, 17, 33, 64};
    int id = 0;
    for (int rep = 0; rep < 6; rep++) for (int n : sizes) {
        CDF* c = (CDF*)malloc(sizeof(CDF) + 64 * sizeof(float));
        c->Reset();
        for (int i = 0; i < n; i++) c->Insert(rep == 0 ? 1.0f : rep == 1 ? (float)(i + 1) : U() * (i % 3 == 0 ? 100.f : 1.f) + 1e-3f);
        printf("cdfw %d %d", id, n);
        for (int i = 0; i < 64; i++) printf(" %u", i < n ? bits(c->data[i]) : 0u);
        printf("\n");
        for (int q = 0; q < 40; q++) {
            float v = U();
            if (q == 0) v = 0.f; if (q == 1) v = 1.f;
            if (q >= 2 && q < 2 + n && q < 12) v = c->data[q - 2] / c->sum;      // exact element boundaries
            if (v > 1.f) v = 1.f;
            unsigned idx = 0; float pdf = 0.f;
            c->Get(v, idx, pdf);
            printf("cdfq %d %u %u %u\n", id, bits(v), idx, bits(pdf));
        }
        free(c);
        id++;
    }
    // ---- make_color rows: rgb bits -> r g b a
    for (int i = 0; i < 3000; i++) {
        float3 c;
        if (i < 600) c = make_float3(U() * 0.01f, U() * 0.0062616f, 0.0031308f + (U() - 0.5f) * 1e-6f);       // around the linear / power switch
        else if (i < 900) c = make_float3(U() * 1.5f - 0.25f, U() * 1.5f - 0.25f, U() * 1.5f - 0.25f);          // outside [0,1]: clamped
        else c = make_float3(U(), U(), U());
        const uchar4 q = make_color(c);
        printf("color %u %u %u %d %d %d %d\n", bits(c.x), bits(c.y), bits(c.z), q.x, q.y, q.z, q.w);
    }
    // ---- binary16 conversion rows: float bits -> half bits -> float bits
    for (int i = 0; i < 4000; i++) {
        float f;
        if (i < 1000) f = (U() * 2.f - 1.f) * 4.f;
        else if (i < 1500) f = (U() * 2.f - 1.f) * 70000.f;                    // overflow to infinity above 65504 + half an ulp
        else if (i < 2000) f = (U() * 2.f - 1.f) * 1.3e-4f;                    // subnormal halves
        else if (i < 2500) f = (U() * 2.f - 1.f) * 1e-7f;                      // underflow to zero / smallest subnormal
        else { uint32_t h = (uint32_t)(rng() % 0x7bffu); __half_raw hr; hr.x = (unsigned short)h; const float a = __half2float(__half(hr));
               hr.x = (unsigned short)(h + 1); const float b = __half2float(__half(hr)); f = 0.5f * a + 0.5f * b; if (i & 1) f = -f; }   // exact ties: round to even
        const __half h = __float2half(f);
        const __half_raw hr = h;
        printf("half %u %u %u\n", bits(f), (unsigned)hr.x, bits(__half2float(h)));
    }
    return 0;
}
