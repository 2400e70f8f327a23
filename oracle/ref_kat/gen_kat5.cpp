// gen_kat5.cpp — known-answer generator, fifth translation unit: the reference's __global__ KERNEL BODIES, compiled from the reference's
// OWN TEXT and run thread by thread on the host (container-only; this file contains no reference source text).
//
// make_kat.py slices these line ranges out of /root/reference into /tmp/lumen_k5_*.inc (never into the repository):
//   ReSTIRKernels.cuh:17-18        RESERVOIR_INDEX / PIXEL_INDEX                         ReSTIRKernels.cuh:29-35    TriangleLightComparator
//   WaveFrontDataStructs.h:13      PIXEL_DATA_INDEX
//   ReSTIRKernels.cu:165-183       CalculateLightWeightsInCDF                            :343-370    FillLightBagsInternal
//   ReSTIRKernels.cu:402-522       PickPrimarySamplesInternal (one token replaced, below) :546-582    GenerateShadowRay
//   ReSTIRKernels.cu:600-616       ShadeInternal                                         :787-980    SpatialNeighbourSamplingInternal
//   ReSTIRKernels.cu:1015-1121     CombineTemporalSamplesInternal                        :1123-1325  CombineUnbiased / CombineBiased / Resample
//   ReSTIRKernels.cu:1407-1436     CombineReservoirBuffersInternal
//   WaveFrontKernels/GPUGeneratePrimRay.cu:8-82 (HaltonSequence + GeneratePrimaryRay), GPUShadeDirect.cu:42-153 (ShadeDirect),
//   GPUShadeIndirect.cu:7-146 (ShadeIndirect)
// The ONE edit: in PickPrimarySamplesInternal the token `__mysmid()` (the hardware SM id, ReSTIRKernels.cu:433: not reproducible even on the
// reference's own hardware) is replaced by `lumen_kat_d2_key(index)` — decision D2 of DESIGN.md: the light bag is keyed on the pixel's
// 16 x 16 tile of the global pixel grid.  Everything else is the reference's text, byte for byte.
//
// What this file supplies is what nvcc supplies: the built-in variables blockIdx / blockDim / threadIdx / gridDim (plain variables, looped over by
// launch1d / launch2d below: one call of the kernel body per thread, in block order), atomicAdd (serial), surf2Dread<ushort2> (reads the motion-vector
// image, a host array), __half22float2 (device-only in the vendored cuda_fp16.h; both halves converted with that header's own host __half2float:
// exact), saturate, and min / max / abs / isnan / isinf from <cmath> / <algorithm>.  ShadeReservoirs (ReSTIRKernels.cu:619-665), the callee that adds a
// reservoir's contribution to the fp16 DIRECT surface with half4 arithmetic (device-only intrinsics; and a non-atomic fp16 read-modify-write this build
// replaces by fp32 accumulation, decision D1) is NOT compiled: a recorder with the declared signature (ReSTIRKernels.cuh) logs every call — which
// reservoir is shaded into which pixel — and the tests price each call as contribution * (weight / 3) in fp32 (since round 6 the callee's own text runs in gen_kat7.cpp, on the
// reference's Half4.h, and the tests bound the distance between that price and its binary16 result per operation).  The OptiX visibility programs
// (WaveFrontShaders.cu:181-216: occluded => reservoir weight = 0) are closed; occlusion comes from a committed pseudo-random mask instead.
// VolumetricShadeDirect (VolumetricKernels/GPUVolumetricShadeDirect.cu:8-101), the first callee of ShadeDirect, is out of scope (volumes): its
// whole body sits under `if (exit T > entry T)` (:23) and draws from the seed only inside; the stand-in below checks that condition is false for every
// pixel (no volume: both are 0) and returns, which is what the reference's function does then.
//
// Rows (every float as the decimal of its bit pattern, integers as decimals), image 64 x 48, three frames f = 0, 1, 2 with a moving camera:
//   prim  i frameCount | x y origin(3) direction(3) contribution(3)                (camera row: camr U(3) V(3) W(3) eye(3))
//   sdir  x y seed surface(40) | emitted origin(3) direction(3) maxDistance radiance(3) channel
//   sind  x y seed surface(40) | emitted origin(3) direction(3) contribution(3)
//   light radiance-sorted list, 16 floats each;  cdfw  per-light weight;  cdf  prefix sums (serial fp32, what CDF::Insert does);
//   seed f a_Seed currentIndex;  surf f pixel surface(40);  mot f pixel ushort2;  occ f pass pixel 0/1
//   bags f i lightIndex pdf (frame 0 only);  res f stage pixel reservoir(17)  stages: 0 pick 1 temporal 2 spatial-1 3 spatial-2 4 combine
//   ray f pass index origin(3) direction(3) distance (in serial append order);  shd f site inX inY outX outY  sites: 0 after pick 1 temporal 2 after spatial
//   surface(40) = flags t position normal tangent incoming transport mat23;  reservoir(17) = weightSum sampleCount weight sample(14);
//   sample(14) = radiance normal position area contribution solidAnglePdf
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cfloat>
#include <random>
#include <vector>
using std::min; using std::max; using std::abs; using std::isnan; using std::isinf;
#include <cuda_runtime.h>
#include <sutil/vec_math.h>
#include <cuda_fp16.h>
#include <nanovdb/NanoVDB.h>

// ---- what nvcc supplies ------------------------------------------------------------------------------------------------------------------
static uint3 blockIdx, threadIdx;
static dim3 blockDim, gridDim;
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned old = *p; *p += v; return old; }
static inline float2 __half22float2(const __half2 h) { return make_float2(__half2float(h.x), __half2float(h.y)); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
struct KatSurface2D { const ushort2* px; unsigned w, h; };
static KatSurface2D g_motionImage;                                       // the one surface object a kernel body reads (handle value 1)
template <class T> static inline void surf2Dread(T* out, cudaSurfaceObject_t, int xBytes, int y, int /*cudaBoundaryModeTrap*/)
{
    static_assert(sizeof(T) == sizeof(ushort2), "only the motion-vector image is read");
    *out = g_motionImage.px[(size_t)y * g_motionImage.w + (size_t)xBytes / sizeof(T)];
}
static unsigned g_katWidth;                                              // D2: bag key = tile of the global 16 x 16 pixel grid
static inline uint32_t lumen_kat_d2_key(int index) { const unsigned y = (unsigned)index / g_katWidth, x = (unsigned)index - y * g_katWidth; return (y / 16u) * ((g_katWidth + 15u) / 16u) + (x / 16u); }

#include "Shaders/CppCommon/MaterialStructs.h"
#include "Shaders/CppCommon/ReSTIRData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/AtomicBuffer.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/IntersectionRayData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/ShadowRayData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/LightData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/SurfaceData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/VolumetricData.h"
#include "Shaders/CppCommon/RenderingUtility.h"
#include "Shaders/CppCommon/Half2.h"
#define lerp lerp_ref
#define __CUDACC__ 1
#include "CUDAKernels/disney.cuh"
using namespace WaveFront;
#include "/tmp/lumen_k5_macros.inc"
#include "/tmp/lumen_k5_pdi.inc"
#include "/tmp/lumen_k5_comparator.inc"
#define CUDA_BLOCK_SIZE 256

// declarations as ReSTIRKernels.cuh:266-296 / GPUVolumetricShadingKernels.cuh:17-25 give them
__device__ __inline__ void CombineUnbiased(Reservoir*, const SurfaceData*, int, Reservoir*, const SurfaceData*, const std::uint32_t);
__device__ __inline__ void CombineBiased(Reservoir*, int, Reservoir*, const SurfaceData*, const std::uint32_t);
__device__ __inline__ void Resample(LightSample*, const SurfaceData*, LightSample*);
static void VolumetricShadeDirect(PixelIndex a_PixelIndex, const uint3 a_ResolutionAndDepth, const VolumetricData* a_VolumetricDataBuffer, AtomicBuffer<ShadowRayData>* const,
                                  const AtomicBuffer<TriangleLight>* const, unsigned int&, const CDF* const = nullptr, cudaSurfaceObject_t = 0)
{
    const auto& v = a_VolumetricDataBuffer[a_PixelIndex.m_Y * a_ResolutionAndDepth.x + a_PixelIndex.m_X];
    if (v.m_ExitIntersectionT > v.m_EntryIntersectionT) { fprintf(stderr, "gen_kat5: a volume in the pixel — out of scope\n"); abort(); }
}
// the recorder standing in for ShadeReservoirs (see the header comment)
struct ShadeCall { unsigned inX, inY, outX, outY; };
static std::vector<ShadeCall> g_shadeCalls;
static void ShadeReservoirs(Reservoir*, unsigned, unsigned a_InputX, unsigned a_InputY, unsigned a_OutputX, unsigned a_OutputY, cudaSurfaceObject_t)
{ g_shadeCalls.push_back({a_InputX, a_InputY, a_OutputX, a_OutputY}); }

#include "/tmp/lumen_k5_restir_fns.inc"          // CombineUnbiased, CombineBiased, Resample
#include "/tmp/lumen_k5_cdfw.inc"
#include "/tmp/lumen_k5_bags.inc"
#include "/tmp/lumen_k5_pick.inc"
#include "/tmp/lumen_k5_genray.inc"
#include "/tmp/lumen_k5_shade.inc"
#include "/tmp/lumen_k5_spatial.inc"
#include "/tmp/lumen_k5_temporal.inc"
#include "/tmp/lumen_k5_combine.inc"
#include "/tmp/lumen_k5_primray.inc"
#include "/tmp/lumen_k5_shadedirect.inc"
#include "/tmp/lumen_k5_shadeindirect.inc"

// ---- launch emulation: one call of the kernel body per thread, blocks in order ----------------------------------------------------------
template <class F> static void launch1d(unsigned numBlocks, unsigned blockSize, F body)
{
    gridDim = dim3(numBlocks, 1, 1); blockDim = dim3(blockSize, 1, 1);
    for (unsigned b = 0; b < numBlocks; b++) for (unsigned t = 0; t < blockSize; t++) { blockIdx = make_uint3(b, 0, 0); threadIdx = make_uint3(t, 0, 0); body(); }
}
template <class F> static void launch2d(unsigned gx, unsigned gy, unsigned bx, unsigned by, F body)
{
    gridDim = dim3(gx, gy, 1); blockDim = dim3(bx, by, 1);
    for (unsigned j = 0; j < gy; j++) for (unsigned i = 0; i < gx; i++) for (unsigned v = 0; v < by; v++) for (unsigned u = 0; u < bx; u++)
    { blockIdx = make_uint3(i, j, 0); threadIdx = make_uint3(u, v, 0); body(); }
}
template <class T> static AtomicBuffer<T>* makeAtomic(unsigned cap)
{
    auto* b = (AtomicBuffer<T>*)calloc(1, sizeof(AtomicBuffer<T>) + sizeof(T) * (size_t)cap);
    b->counter = 0; b->maxSize = cap; return b;
}

// ---- synthetic inputs ------------------------------------------------------------------------------------------------------------------
static std::mt19937 rng(20261005u);
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(rng); }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float3 unitvec() { for (;;) { float3 v = make_float3(U()*2-1, U()*2-1, U()*2-1); float l = length(v); if (l > 0.1f && l <= 1.f) return v / l; } }
struct MatIn { float c[4], tint[3], lum, trn[3], ior, p[11]; };      // p: metallic subsurface specular roughness spectint anisotropic sheen sheentint clearcoat clearcoatgloss transmission
static MaterialData build(const MatIn& in)
{
    MaterialData m(0.f);
    m.SetColor(make_float4(in.c[0], in.c[1], in.c[2], in.c[3]));
    m.SetTint(make_float3(in.tint[0], in.tint[1], in.tint[2]));
    m.SetLuminance(in.lum);
    m.SetTransmittance(make_float3(in.trn[0], in.trn[1], in.trn[2]));
    m.SetRefractiveIndex(in.ior);
    m.SetMetallic(in.p[0]); m.SetSubSurface(in.p[1]); m.SetSpecular(in.p[2]); m.SetRoughness(in.p[3]);
    m.SetSpecTint(in.p[4]); m.SetAnisotropic(in.p[5]); m.SetSheen(in.p[6]); m.SetSheenTint(in.p[7]);
    m.SetClearCoat(in.p[8]); m.SetClearCoatGloss(in.p[9]); m.SetTransmission(in.p[10]);
    return m;
}
static MatIn randmat(int kind)
{
    MatIn in{};
    for (int i = 0; i < 3; i++) { in.c[i] = 0.1f + 0.9f * U(); in.tint[i] = U(); in.trn[i] = U() * 2.f; }
    in.c[3] = 1.f; in.lum = 0.25f + U();
    in.ior = (kind & 1) ? 1.f / (1.1f + U()) : 1.f;
    for (int i = 0; i < 11; i++) in.p[i] = 0.f;
    in.p[3] = 0.05f + 0.95f * U();
    switch (kind % 6) {
    case 0: break;
    case 1: in.p[0] = U(); in.p[2] = U(); in.p[4] = U(); break;
    case 2: in.p[0] = U(); in.p[2] = U(); in.p[6] = U(); in.p[7] = U(); in.p[1] = U(); break;
    case 3: in.p[0] = U(); in.p[2] = U(); in.p[8] = U(); in.p[9] = U(); break;
    case 4: in.p[0] = U()*0.5f; in.p[2] = U(); in.p[5] = U(); in.p[10] = 0.2f + 0.8f*U(); in.ior = 1.f/(1.1f+U()); break;
    case 5: for (int i = 0; i < 11; i++) if (i != 3) in.p[i] = U(); in.ior = 1.f/(1.1f+U()); break;
    }
    return in;
}
static const unsigned W = 64, H = 48, N = W * H;
// the "scene" behind the synthetic G-buffer: floor y = 0, back wall z = -3 for |x| < 2.7 (beside it: miss), a sphere; materials by object and 8 x 8 checker
static MatIn g_mats[12];
struct Cam { float3 eye; };
static float3 camDir(const Cam&, unsigned x, unsigned y, float jx, float jy)
{
    const float u = ((float)x + jx) / (float)W * 2.f - 1.f, v = 1.f - ((float)y + jy) / (float)H * 2.f;
    return normalize(make_float3(u * (float)W / (float)H, v, -2.4f));
}
static bool project(const Cam& c, const float3& p, float& sx, float& sy)
{
    const float3 d = p - c.eye;
    if (d.z >= -1e-3f) return false;
    const float k = -2.4f / d.z;
    sx = ((d.x * k) / ((float)W / (float)H) + 1.f) * 0.5f; sy = (1.f - d.y * k) * 0.5f;
    return true;
}
static void makeSurface(const Cam& cam, unsigned x, unsigned y, SurfaceData& s, MatIn& matOut)
{
    memset(&s, 0, sizeof s);
    s.m_PixelIndex = PixelIndex{(unsigned short)x, (unsigned short)y};
    const float3 o = cam.eye, d = camDir(cam, x, y, U(), U());
    float best = 1e30f; int obj = -1; float3 n = make_float3(0, 0, 0);
    if (d.y < -1e-4f) { const float t = -o.y / d.y; if (t > 0.f && t < best) { best = t; obj = 0; n = make_float3(0, 1, 0); } }
    if (d.z < -1e-4f) { const float t = (-3.f - o.z) / d.z; const float hy = o.y + d.y * t; if (t > 0.f && t < best && hy >= 0.f && fabsf(o.x + d.x * t) < 2.7f) { best = t; obj = 1; n = make_float3(0, 0, 1); } }
    { const float3 c = make_float3(0.3f, 0.8f, -1.f); const float r = 0.8f; const float3 oc = o - c; const float b = dot(oc, d), cc = dot(oc, oc) - r * r, disc = b * b - cc;
      if (disc > 0.f) { const float t = -b - sqrtf(disc); if (t > 0.f && t < best) { best = t; obj = 2; n = normalize(o + d * t - c); } } }
    matOut = g_mats[0];
    if (obj < 0) { s.m_SurfaceFlags = SURFACE_FLAG_NON_INTERSECT; return; }                       // GPUExtractSurfaceData.cu:222-226: only the flag is set
    const float3 p = o + d * best;
    const int checker = ((int)floorf(p.x * 2.f) + (int)floorf((obj == 1 ? p.y : p.z) * 2.f)) & 1;
    const int mi = obj * 4 + checker * 2 + ((x / 24) & 1);
    matOut = g_mats[mi % 12];
    s.m_IntersectionT = best;
    // shading normal: the geometric one, perturbed a little (normal mapping), per pixel
    const float3 ns = normalize(n + make_float3(U() - 0.5f, U() - 0.5f, U() - 0.5f) * 0.08f);
    s.m_Normal = ns;
    if (obj == 1 && fabsf(p.x + 1.2f) < 0.35f && fabsf(p.y - 1.5f) < 0.3f) {                         // an emitter seen directly: colour normalised, nothing else (:120-136)
        s.m_SurfaceFlags = SURFACE_FLAG_EMISSIVE; s.m_MaterialData.m_Color = make_float4(1.f, 0.8f, 0.5f, 1.f); return;
    }
    s.m_Position = p; s.m_IncomingRayDirection = d; s.m_TransportFactor = make_float3(1.f, 1.f, 1.f);
    if (obj == 0 && fabsf(p.x - 1.0f) < 0.3f && fabsf(p.z + 0.2f) < 0.4f) { s.m_SurfaceFlags = SURFACE_FLAG_ALPHA_TRANSPARENT; return; }    // alpha cut-out (:139-151)
    s.m_GeometricNormal = n;
    float3 t = unitvec(); t = t - ns * dot(t, ns);
    s.m_Tangent = normalize(t);
    s.m_MaterialData = build(matOut);
}
static void printMat(const MatIn& in)
{
    for (float v : in.c) printf(" %u", bits(v)); for (float v : in.tint) printf(" %u", bits(v)); printf(" %u", bits(in.lum));
    for (float v : in.trn) printf(" %u", bits(v)); printf(" %u", bits(in.ior)); for (float v : in.p) printf(" %u", bits(v));
}
static void print3(const float3& v) { printf(" %u %u %u", bits(v.x), bits(v.y), bits(v.z)); }
static void printSurface(const SurfaceData& s, const MatIn& m)
{
    printf(" %u %u", (unsigned)s.m_SurfaceFlags, bits(s.m_IntersectionT));
    print3(s.m_Position); print3(s.m_Normal); print3(s.m_Tangent); print3(s.m_IncomingRayDirection); print3(s.m_TransportFactor);
    // a flagged surface carries no material parameters; the emissive one carries its colour in the first four floats
    if (s.m_SurfaceFlags & SURFACE_FLAG_EMISSIVE) { MatIn e{}; e.c[0] = s.m_MaterialData.m_Color.x; e.c[1] = s.m_MaterialData.m_Color.y; e.c[2] = s.m_MaterialData.m_Color.z; e.c[3] = s.m_MaterialData.m_Color.w; printMat(e); }
    else if (s.m_SurfaceFlags) { MatIn e{}; printMat(e); }
    else printMat(m);
}
static void printSample(const LightSample& l)
{
    print3(l.radiance); print3(l.normal); print3(l.position); printf(" %u", bits(l.area)); print3(l.unshadowedPathContribution); printf(" %u", bits(l.solidAnglePdf));
}
// A reservoir that never took a sample leaves a kernel with its LightSample as constructed (ReSTIRData.h:97): everything zero EXCEPT unshadowedPathContribution, which
// the constructor does not initialise — stack contents, under ASLR not even reproducible by this generator.  Recognisable by area == 0 (every light has area > 0).
// Those three cells are set to 0 in the buffer itself (the value this build's zero-initialised reservoirs hold, decision D5) before anything reads or prints them.
static void printRes(const char* tag, int f, int stage, Reservoir* r)
{
    for (unsigned i = 0; i < N; i++) if (r[i].sample.area == 0.f) r[i].sample.unshadowedPathContribution = make_float3(0.f, 0.f, 0.f);
    for (unsigned i = 0; i < N; i++) { printf("%s %d %d %u %u %lld %u", tag, f, stage, i, bits(r[i].weightSum), r[i].sampleCount, bits(r[i].weight)); printSample(r[i].sample); printf("\n"); }
}

int main()
{
    g_katWidth = W;
    for (int i = 0; i < 12; i++) g_mats[i] = randmat(i < 8 ? (i % 3) : i);      // mostly the opaque stack, a few clear-coat / transmission / everything-on surfaces

    // ---- GeneratePrimaryRay (GPUGeneratePrimRay.cu:28-82)
    {
        const float3 Uc = make_float3(-1.31f, 0.02f, 0.11f), Vc = make_float3(0.03f, 0.97f, -0.05f), Wc = make_float3(0.08f, -0.04f, -0.99f), eye = make_float3(0.4f, 1.3f, 3.9f);
        printf("camr"); print3(Uc); print3(Vc); print3(Wc); print3(eye); printf("\n");
        auto* rays = makeAtomic<IntersectionRayData>(N);
        for (unsigned frameCount : {1u, 3u, 12345u, 0xfffffff0u}) {
            launch1d((N + 255u) / 256u, 256u, [&] { GeneratePrimaryRay((int)N, rays, Uc, Vc, Wc, eye, make_uint2(W, H), frameCount, 0); });
            for (unsigned i = 0; i < N; i++) {
                const auto& r = rays->data[i];
                printf("prim %u %u %u %u", i, frameCount, (unsigned)r.m_PixelIndex.m_X, (unsigned)r.m_PixelIndex.m_Y); print3(r.m_Origin); print3(r.m_Direction); print3(r.m_Contribution); printf("\n");
            }
        }
        free(rays);
    }

    // ---- lights: sorted by the reference's comparator (distinct keys: the order does not depend on the sort's stability), weights by its kernel, serial prefix sums
    const unsigned L = 37;
    auto* lights = makeAtomic<TriangleLight>(L);
    for (unsigned i = 0; i < L; i++) {
        TriangleLight t;
        const float3 c = make_float3(U() * 5.f - 2.5f, 0.3f + U() * 2.6f, U() * 4.f - 2.5f);
        t.p0 = c; t.p1 = c + unitvec() * (0.1f + 0.5f * U()); t.p2 = c + unitvec() * (0.1f + 0.5f * U());
        const float3 cr = cross(t.p1 - t.p0, t.p2 - t.p0);
        t.normal = normalize(cr); t.area = 0.5f * length(cr);
        t.radiance = make_float3(U() * 30.f, U() * 30.f, U() * 30.f) * (i % 5 == 0 ? 0.05f : 1.f);
        lights->data[i] = t;
    }
    lights->counter = L;
    std::sort(lights->data, lights->data + L, TriangleLightComparator());
    CDF* cdf = (CDF*)calloc(1, sizeof(CDF) + sizeof(float) * L);
    launch1d((L + 511u) / 512u, 512u, [&] { CalculateLightWeightsInCDF(cdf, lights, L); });
    for (unsigned i = 0; i < L; i++) {
        const auto& t = lights->data[i];
        printf("light"); print3(t.p0); print3(t.p1); print3(t.p2); print3(t.normal); print3(t.radiance); printf(" %u\n", bits(t.area));
        printf("cdfw %u\n", bits(cdf->data[i]));
    }
    { float acc = 0.f; for (unsigned i = 0; i < L; i++) { acc += cdf->data[i]; cdf->data[i] = acc; } }      // thrust::inclusive_scan (ReSTIRKernels.cu:88-90): order unspecified; serial = CDF::Insert
    cdf->SetCDFSize(L);                                                                                     // SetCDFSize kernel (:185-190) = this member
    for (unsigned i = 0; i < L; i++) printf("cdf %u\n", bits(cdf->data[i]));

    // ---- the ReSTIR chain of Framework/ReSTIR.cpp:65-233 over three frames
    std::vector<SurfaceData> surf[2] = {std::vector<SurfaceData>(N), std::vector<SurfaceData>(N)};
    { SurfaceData z; memset(&z, 0, sizeof z); for (auto& v : surf) std::fill(v.begin(), v.end(), z); }      // WaveFrontRenderer.cpp:652: memset before the first frame
    std::vector<Reservoir> res[4];
    for (auto& v : res) { v.resize(N); for (auto& r : v) { memset(&r, 0, sizeof r); } }                       // ResetReservoirs (:36-47) on zeroed memory
    auto* bags = (LightBagEntry*)calloc(50u * 1000u, sizeof(LightBagEntry));
    auto* rays = makeAtomic<RestirShadowRay>(N);
    std::vector<ushort2> motion(N);
    int swapIndex = 0, frameIndex = 0;
    Cam prevCam{make_float3(0.f, 1.f, 4.f)};
    for (int f = 0; f < 3; f++) {
        const Cam cam{make_float3(0.06f * (float)f, 1.f + 0.02f * (float)f, 4.f - 0.05f * (float)f)};
        const int curS = frameIndex, prevS = frameIndex ^ 1;
        std::vector<MatIn> mats(N);
        for (unsigned y = 0; y < H; y++) for (unsigned x = 0; x < W; x++) makeSurface(cam, x, y, surf[curS][y * W + x], mats[y * W + x]);
        // motion vectors: where the surface point was on the previous frame's screen, minus where it is now, as binary16 (MotionVectors.cu:8-55 stores half2)
        for (unsigned i = 0; i < N; i++) {
            const SurfaceData& s = surf[curS][i];
            float mx = 0.f, my = 0.f;
            if (s.m_IntersectionT > 0.f && !(s.m_SurfaceFlags & SURFACE_FLAG_EMISSIVE)) {
                float px, py;
                if (project(prevCam, s.m_Position, px, py)) { mx = px - ((float)(i % W) + 0.5f) / (float)W; my = py - ((float)(i / W) + 0.5f) / (float)H; }
            }
            if (i % 97 == 5) mx = 2.f;                                                                      // out of the image: falls back to the same pixel (:1044-1061)
            const __half hx = __float2half(mx), hy = __float2half(my);
            memcpy(&motion[i].x, &hx, 2); memcpy(&motion[i].y, &hy, 2);
        }
        g_motionImage = {motion.data(), W, H};
        const uint32_t a_Seed = rng();
        const int currentIndex = swapIndex, temporalIndex = currentIndex == 1 ? 0 : 1;
        printf("seed %d %u %d\n", f, a_Seed, currentIndex);
        for (unsigned i = 0; i < N; i++) { printf("surf %d %u", f, i); printSurface(surf[curS][i], mats[i]); printf("\n"); printf("mot %d %u %u %u\n", f, i, (unsigned)motion[i].x, (unsigned)motion[i].y); }
        Reservoir* RC = res[currentIndex].data(); Reservoir* RT = res[temporalIndex].data();
        const SurfaceData* cur = surf[curS].data(); const SurfaceData* prev = surf[prevS].data();
        const uint2 dims = make_uint2(W, H);

        uint32_t seed = WangHash(a_Seed);
        launch1d((50u * 1000u + 255u) / 256u, 256u, [&] { FillLightBagsInternal(50u, 1000u, cdf, bags, lights, a_Seed); });
        if (f == 0) for (unsigned i = 0; i < 50000u; i++) {
            unsigned li = 0; while (li < L && memcmp(&lights->data[li], &bags[i].light, sizeof(TriangleLight)) != 0) ++li;
            printf("bags %d %u %u %u\n", f, i, li, bits(bags[i].pdf));
        }
        seed = WangHash(seed);
        launch1d((N + 255u) / 256u, 256u, [&] { PickPrimarySamplesInternal(bags, RC, 32u, N, 50u, 1000u, cur, seed); });
        printRes("res", f, 0, RC);
        auto visibility = [&](int pass) {
            rays->counter = 0;
            launch1d((N + 255u) / 256u, 256u, [&] { GenerateShadowRay(rays, RC, cur, N); });
            std::vector<unsigned char> occ(N);
            for (unsigned i = 0; i < N; i++) occ[i] = (rng() % 100u) < 35u;                                 // 35 % of the rays are blocked
            for (unsigned i = 0; i < N; i++) printf("occ %d %d %u %u\n", f, pass, i, (unsigned)occ[i]);
            for (unsigned k = 0; k < rays->counter; k++) {
                const RestirShadowRay& r = rays->data[k];
                printf("ray %d %d %u", f, pass, r.index); print3(r.origin); print3(r.direction); printf(" %u\n", bits(r.distance));
                if (occ[r.index]) RC[r.index].weight = 0.f;                                                 // __anyhit__ of the ReSTIR ray type (WaveFrontShaders.cu:197-210)
            }
        };
        auto shadeAll = [&](int site) {
            g_shadeCalls.clear();
            launch2d((W + 31u) / 32u, (H + 31u) / 32u, 32u, 32u, [&] { ShadeInternal(RC, W, H, 0); });
            for (const auto& c : g_shadeCalls) printf("shd %d %d %u %u %u %u\n", f, site, c.inX, c.inY, c.outX, c.outY);
        };
        visibility(0);
        shadeAll(0);
        seed = WangHash(seed);
        g_shadeCalls.clear();
        launch1d((N + 255u) / 256u, 256u, [&] { CombineTemporalSamplesInternal(RC, RT, cur, prev, seed, N, dims, 1, 0); });
        for (const auto& c : g_shadeCalls) printf("shd %d %d %u %u %u %u\n", f, 1, c.inX, c.inY, c.outX, c.outY);
        printRes("res", f, 1, RC);
        seed = WangHash(seed);
        launch1d((N + 255u) / 256u, 256u, [&] { SpatialNeighbourSamplingInternal(RC, res[2].data(), cur, seed, dims, N); });
        printRes("res", f, 2, res[2].data());
        launch1d((N + 255u) / 256u, 256u, [&] { SpatialNeighbourSamplingInternal(res[2].data(), res[3].data(), cur, seed, dims, N); });
        printRes("res", f, 3, res[3].data());
        visibility(1);
        shadeAll(2);
        launch1d((N + 255u) / 256u, 256u, [&] { CombineReservoirBuffersInternal(RC, res[3].data(), cur, N, WangHash(seed)); });
        printRes("res", f, 4, RC);
        // one wave per frame here (WaveFrontRenderer.cpp:827 swaps per executed wave): the chain turns, the history is live in the next frame
        swapIndex = swapIndex + 1 >= 2 ? 0 : swapIndex + 1;
        frameIndex ^= 1;
        prevCam = cam;
    }

    // ---- ShadeDirect / ShadeIndirect (GPUShadeDirect.cu:42-153, GPUShadeIndirect.cu:7-146) on the last frame's surfaces with path-like transport factors
    {
        std::vector<SurfaceData> sd(surf[frameIndex ^ 1]);
        std::vector<MatIn> mats(N);
        Cam cam{make_float3(0.12f, 1.04f, 3.9f)};
        for (unsigned y = 0; y < H; y++) for (unsigned x = 0; x < W; x++) {
            makeSurface(cam, x, y, sd[y * W + x], mats[y * W + x]);
            SurfaceData& s = sd[y * W + x];
            if (!(s.m_SurfaceFlags & (SURFACE_FLAG_EMISSIVE | SURFACE_FLAG_NON_INTERSECT))) s.m_TransportFactor = make_float3(U(), U(), U());
            if ((x + y) % 11 == 0 && !s.m_SurfaceFlags) s.m_IncomingRayDirection = normalize(s.m_IncomingRayDirection - s.m_Normal * (dot(s.m_IncomingRayDirection, s.m_Normal) * (1.f - 1e-4f * U())));   // grazing: |dot| < 3e-4 for some
        }
        std::vector<VolumetricData> vol(N); memset(vol.data(), 0, sizeof(VolumetricData) * N);
        auto* shadow = makeAtomic<ShadowRayData>(N); auto* volShadow = makeAtomic<ShadowRayData>(5 * N); auto* out = makeAtomic<IntersectionRayData>(N);
        for (uint32_t a_Seed : {WangHash(7u), WangHash(WangHash(7u))}) {
            shadow->counter = 0; volShadow->counter = 0; out->counter = 0;
            launch2d((W + 15u) / 16u, (H + 15u) / 16u, 16u, 16u, [&] { ShadeDirect(make_uint3(W, H, 2), sd.data(), vol.data(), lights, a_Seed, cdf, shadow, volShadow, 0); });
            launch2d((W + 15u) / 16u, (H + 15u) / 16u, 16u, 16u, [&] { ShadeIndirect(make_uint3(W, H, 2), sd.data(), out, a_Seed); });
            if (volShadow->counter) { fprintf(stderr, "volumetric shadow rays without a volume\n"); return 1; }
            std::vector<int> sAt(N, -1), iAt(N, -1);
            for (unsigned k = 0; k < shadow->counter; k++) sAt[shadow->data[k].m_PixelIndex.m_Y * W + shadow->data[k].m_PixelIndex.m_X] = (int)k;
            for (unsigned k = 0; k < out->counter; k++) iAt[out->data[k].m_PixelIndex.m_Y * W + out->data[k].m_PixelIndex.m_X] = (int)k;
            for (unsigned i = 0; i < N; i++) {
                printf("sdir %u %u %u", i % W, i / W, a_Seed); printSurface(sd[i], mats[i]);
                if (sAt[i] >= 0) { const auto& r = shadow->data[sAt[i]]; printf(" 1"); print3(r.m_Origin); print3(r.m_Direction); printf(" %u", bits(r.m_MaxDistance)); print3(r.m_PotentialRadiance); printf(" %u\n", (unsigned)r.m_OutputChannel); }
                else printf(" 0 0 0 0 0 0 0 0 0 0 0 0\n");
                printf("sind %u %u %u", i % W, i / W, a_Seed); printSurface(sd[i], mats[i]);
                if (iAt[i] >= 0) { const auto& r = out->data[iAt[i]]; printf(" 1"); print3(r.m_Origin); print3(r.m_Direction); print3(r.m_Contribution); printf("\n"); }
                else printf(" 0 0 0 0 0 0 0 0 0 0\n");
            }
        }
    }
    return 0;
}
