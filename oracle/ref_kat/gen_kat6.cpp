// gen_kat6.cpp — known-answer generator, sixth translation unit: the reference's SCENE-FACING kernel bodies, compiled from the reference's own text and run thread by
// thread on the host (container-only; contains no reference source text).  make_kat.py slices into /tmp/lumen_k6_*.inc:
//   WaveFrontKernels/GPUExtractSurfaceData.cu:8-228   ExtractSurfaceDataGpu        (hit record + ray -> SurfaceData: the G-buffer of depth 0 and every deeper wave)
//   MotionVectors.cu:8-55                             GenerateMotionVector
//   WaveFrontKernels/GPUShadeDirect.cu:11-40          ResolveDirectLightHits (rows xres k half4 bits: the DIRECT channel after the kernel, cleared before)
//   WaveFrontKernels/GPUDataBufferKernels.cu:9-186    BuildLightDataBufferGPU + BuildLightDataInstance   (the per-frame emissive-triangle list)
//   WaveFrontKernels/GPUEmissiveLookup.cu:13-109      FindEmissivesGpu             (per-primitive emissive flags at load time)
//   WaveFrontDataStructs.h:13 (PIXEL_DATA_INDEX), and Shaders/CppCommon/WaveFrontDataStructs/IntersectionData.h whole, with ONE token changed: the default
//   constructor's `m_Barycentrics({0.f, 0.f})` (:33) is ambiguous against the host-side constructors of the vendored __half2 (nvcc's device view has fewer);
//   it becomes `m_Barycentrics()` — the harness never default-constructs a hit record.
// Supplied here, as nvcc would: blockIdx / blockDim / threadIdx / gridDim, atomicAdd (serial), __float22half2_rn (device-only in the vendored cuda_fp16.h; both
// halves with that header's host __float2half_rn), surf2Dwrite<ushort2> (into a host image), and tex2D<float4>.  THE TEXTURE UNIT IS HARDWARE: it is pinned only
// where no filtering can happen — every texture of this scene is 1 x 1, so any fetch returns the one texel whatever the filter and address mode; a texel is the
// float4 cudaReadModeNormalizedFloat makes of RGBA8 (u8 / 255.f), none sRGB-flagged (the flag is a hardware decode as well).  A null texture object (the
// clear-coat-roughness slot, quirk 12) returns zeros (decision D5).  Host-side code of the reference that feeds these kernels (PTMaterial setters, SceneDataTable,
// LightDataBuffer's instance list, the launch shapes of CPUDataBufferKernels.cu:36-56) cannot run here (it allocates CUDA memory); the harness builds the same
// tables by hand and cites the lines.
// Rows (floats as bit patterns):
//   xtex t r g b a                      1 x 1 textures
//   xmat m color4 emission3 tint3 lum transmittance3 ior p11 | tex ids: diffuse normal metalRough emissive transmission clearCoat clearCoatRough tint
//   xvert p v pos3 uv2 normal3 tangent4;  xidx p i0 i1 i2;  xprim p material numLights(FindEmissivesGpu);  xemis p tri flag
//   xinst i prim mode transform16 radiance3 scale          (table entry i = instance i: one primitive per mesh)
//   xmvm 16 floats (projection * inverse(previous camera), row major);  xeye 3 floats
//   xhit k set entry prim baryU baryV(binary16 bits) t px py dir3 contribution3
//   xsurf k set | flags t position normal geomNormal tangent incoming transport color4 tint4 transmittance4 params3        (35 words)
//   xmv k ushort2 (set 0 only);   xlight k 16 floats (slot order of the light buffer, reserved-but-unset slots included);  xnlights total
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
#include <sutil/Matrix.h>
#include <cuda_fp16.h>

static uint3 blockIdx, threadIdx;
static dim3 blockDim, gridDim;
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned old = *p; *p += v; return old; }
static inline __half2 __float22half2_rn(const float2 f) { __half2 h; h.x = __float2half_rn(f.x); h.y = __float2half_rn(f.y); return h; }
static inline float2 __half22float2(const __half2 h) { return make_float2(__half2float(h.x), __half2float(h.y)); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
static std::vector<ushort2> g_mvImage; static unsigned g_mvWidth;
static std::vector<ushort4> g_directImage;                // the DIRECT light channel (a half4 surface) as ResolveDirectLightHits writes it
template <class T> static inline void surf2Dwrite(T v, cudaSurfaceObject_t, int xBytes, int y, int)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "motion vectors (ushort2) and half4 pixels (ushort4) only");
    if (sizeof(T) == 4) memcpy(&g_mvImage[(size_t)y * g_mvWidth + (size_t)xBytes / 4], &v, 4);
    else memcpy(&g_directImage[(size_t)y * g_mvWidth + (size_t)xBytes / 8], &v, 8);
}
// Half4.h does not compile on the host (its arithmetic is device intrinsics); ResolveDirectLightHits only CONSTRUCTS a half4 from a float4 and stores its bits:
// the same two members filled by the same per-element conversion as Half4.h:34-38 (the vendored header's host __float2half)
struct half4 { __half2 m_Elements[2]; half4(const float4& f) { m_Elements[0].x = __float2half(f.x); m_Elements[0].y = __float2half(f.y); m_Elements[1].x = __float2half(f.z); m_Elements[1].y = __float2half(f.w); } };
union half4Ushort4 { half4 m_Half4; ushort4 m_Ushort4; half4Ushort4(const float4& f) : m_Half4(f) {} };
template <class T> static inline T tex2D(cudaTextureObject_t t, float, float) { static_assert(sizeof(T) == 16, "float4 fetches only"); if (!t) return T{0.f, 0.f, 0.f, 0.f}; return *reinterpret_cast<const T*>(t); }

#include "Shaders/CppCommon/MaterialStructs.h"
#include "Shaders/CppCommon/ModelStructs.h"
#include "Shaders/CppCommon/SceneDataTableAccessor.h"
#include <Lumen/ModelLoading/MeshInstance.h>
#include "Shaders/CppCommon/WaveFrontDataStructs/AtomicBuffer.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/IntersectionRayData.h"
#include "/tmp/lumen_k6_intersectiondata.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/LightData.h"
#include "Shaders/CppCommon/WaveFrontDataStructs/SurfaceData.h"
#include "Shaders/CppCommon/Half2.h"
#include "Framework/LightDataBuffer.h"
#define lerp lerp_ref
#define __CUDACC__ 1
#include "CUDAKernels/disney.cuh"
using namespace WaveFront;
#include "/tmp/lumen_k6_pdi.inc"
GPU_ONLY void BuildLightDataInstance(const LightInstanceData&, const SceneDataTableAccessor*, uint32_t, uint32_t, uint32_t, WaveFront::AtomicBuffer<WaveFront::TriangleLight>*);
#include "/tmp/lumen_k6_extract.inc"
#include "/tmp/lumen_k6_motion.inc"
#include "/tmp/lumen_k6_lights.inc"
#include "/tmp/lumen_k6_emissives.inc"
#include "/tmp/lumen_k6_resolve.inc"

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
static std::mt19937 rng(20261006u);
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(rng); }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float3 unitvec() { for (;;) { const float a = U()*2-1, b = U()*2-1, c = U()*2-1; float3 v = make_float3(a, b, c); float l = length(v); if (l > 0.1f && l <= 1.f) return v / l; } }
static void p3(const float3& v) { printf(" %u %u %u", bits(v.x), bits(v.y), bits(v.z)); }
static void p4(const float4& v) { printf(" %u %u %u %u", bits(v.x), bits(v.y), bits(v.z), bits(v.w)); }
static float mid8() { return ((float)(rng() % 255u) + 0.5f) / 255.f; }      // a parameter in the middle of its 8-bit bucket: an ulp of noise upstream cannot move it

static const unsigned W = 64, H = 48, N = W * H;

int main()
{
    // ---- 1 x 1 textures: 0 white, 1 default normal (128,128,255,0), then random ones (alpha of the base-colour candidates on both sides of the 0.51 cut-out)
    const unsigned NT = 14;
    uchar4 tex8[NT]; float4 texel[NT];
    tex8[0] = uchar4{255, 255, 255, 255}; tex8[1] = uchar4{128, 128, 255, 0};
    for (unsigned t = 2; t < NT; t++) tex8[t] = uchar4{(unsigned char)(rng() % 256u), (unsigned char)(rng() % 256u), (unsigned char)(rng() % 256u), (unsigned char)(t % 3 == 0 ? rng() % 120u : 180u + rng() % 76u)};
    for (unsigned t = 0; t < NT; t++) {
        texel[t] = make_float4((float)tex8[t].x / 255.f, (float)tex8[t].y / 255.f, (float)tex8[t].z / 255.f, (float)tex8[t].w / 255.f);
        printf("xtex %u %u %u %u %u\n", t, (unsigned)tex8[t].x, (unsigned)tex8[t].y, (unsigned)tex8[t].z, (unsigned)tex8[t].w);
    }
    auto handle = [&](int t) -> cudaTextureObject_t { return t < 0 ? 0ull : (cudaTextureObject_t)(uintptr_t)&texel[t]; };

    // ---- materials.  MaterialData as WaveFrontRenderer::CreateMaterial leaves it through the PTMaterial setters (WaveFrontRenderer.cpp:1269-1318, PTMaterial.cpp:10-19,38-200):
    // MaterialData(0), then colour, emission, the factor setters; DeviceMaterial as PTMaterial::CreateDeviceMaterial fills it (PTMaterial.cpp:97-148, quirk 12)
    const unsigned NM = 8;
    std::vector<DeviceMaterial> mats(NM);
    for (unsigned m = 0; m < NM; m++) {
        float c[4] = {0.1f + 0.9f * U(), 0.1f + 0.9f * U(), 0.1f + 0.9f * U(), 0.75f + 0.25f * U()};          // alpha: the cut-out (< 0.51 after the texture) is decided by the texture
        float e[3] = {0.f, 0.f, 0.f};
        if (m == 1) { e[0] = 3.f * U(); e[1] = 3.f * U(); e[2] = 3.f * U(); }          // an emissive material (EmissionMode::ENABLED instances light up)
        float tint[3] = {U(), U(), U()}, trn[3] = {2.f * U(), 2.f * U(), 2.f * U()};
        const float lum = 0.25f + U(), ior = 1.f + 0.9f * U();
        float p[11];
        for (float& v : p) v = mid8();
        if (m % 2 == 0) { p[8] = 0.f; p[10] = 0.f; }                                     // half of them without clear coat / transmission
        int ids[8] = {(int)(2 + m % (NT - 2)), m % 2 ? 1 : (int)(2 + (m * 3) % (NT - 2)), (int)(2 + (m * 5) % (NT - 2)), m == 1 ? (int)(2 + (m * 7) % (NT - 2)) : 0,
                      m % 3 ? 0 : (int)(2 + (m * 2) % (NT - 2)), m % 3 == 1 ? (int)(2 + m) : 0, m % 2 ? (int)(3 + m) : 0, m % 2 ? 0 : (int)(4 + m)};
        MaterialData d(0.f);
        d.SetColor(make_float4(c[0], c[1], c[2], c[3]));
        d.SetEmissive(make_float3(e[0], e[1], e[2]));
        d.SetTransmission(p[10]); d.SetClearCoat(p[8]); d.SetClearCoatGloss(p[9]); d.SetRefractiveIndex(ior);
        d.SetSpecular(p[2]); d.SetSpecTint(p[4]); d.SetSubSurface(p[1]); d.SetLuminance(lum); d.SetAnisotropic(p[5]); d.SetSheen(p[6]); d.SetSheenTint(p[7]);
        d.SetTint(make_float3(tint[0], tint[1], tint[2])); d.SetTransmittance(make_float3(trn[0], trn[1], trn[2]));
        d.SetRoughness(p[3]); d.SetMetallic(p[0]);
        DeviceMaterial& dm = mats[m];
        dm.m_MaterialData = d;
        dm.m_DiffuseTexture = handle(ids[0]); dm.m_NormalTexture = handle(ids[1]); dm.m_MetalRoughnessTexture = handle(ids[2]); dm.m_EmissiveTexture = handle(ids[3]);
        dm.m_TransmissionTexture = handle(ids[4]); dm.m_TintTexture = handle(ids[7]);
        dm.m_ClearCoatTexture = handle(ids[6]);           // sic: the clear-coat-ROUGHNESS texture lands in the clear-coat slot (PTMaterial.cpp:126-129) ...
        dm.m_ClearCoatRoughnessTexture = 0;               // ... and the roughness slot keeps the constructor's null handle
        printf("xmat %u", m); for (float v : c) printf(" %u", bits(v)); for (float v : e) printf(" %u", bits(v)); for (float v : tint) printf(" %u", bits(v));
        printf(" %u", bits(lum)); for (float v : trn) printf(" %u", bits(v)); printf(" %u", bits(ior)); for (float v : p) printf(" %u", bits(v));
        for (int v : ids) printf(" %d", v);
        printf("\n");
    }
    // ---- primitives: triangle soups with per-vertex attributes; FindEmissivesGpu at load time (CPUDataBufferKernels.cu:6-20: one thread)
    const unsigned NP = 8;                                                              // one per material
    struct Prim { std::vector<Vertex> v; std::vector<uint32_t> idx; std::vector<unsigned char> emis; unsigned mat, numLights; };
    std::vector<Prim> prims(NP);
    for (unsigned p = 0; p < NP; p++) {
        Prim& pr = prims[p];
        const unsigned nv = 12 + 3 * p, nt = 9 + 2 * p;
        pr.mat = p;
        pr.v.resize(nv);
        memset(pr.v.data(), 0, nv * sizeof(Vertex));
        for (unsigned k = 0; k < nv; k++) {
            Vertex& v = pr.v[k];
            const float px = U() * 4.f - 2.f, py = U() * 4.f - 2.f, pz = U() * 4.f - 2.f, uu = U() * 3.f - 1.f, vv = U() * 3.f - 1.f;
            v.m_Position = make_float3(px, py, pz); v.m_UVCoord = make_float2(uu, vv);
            const float3 n = unitvec(); float3 t = unitvec(); t = normalize(t - n * dot(t, n));
            v.m_Normal = n; v.m_Tangent = make_float4(t.x, t.y, t.z, (k % 3) ? 1.f : -1.f);
            printf("xvert %u %u", p, k); p3(v.m_Position); printf(" %u %u", bits(v.m_UVCoord.x), bits(v.m_UVCoord.y)); p3(v.m_Normal); p4(v.m_Tangent); printf("\n");
        }
        for (unsigned t = 0; t < nt; t++) {
            unsigned a = rng() % nv, b = rng() % nv, c = rng() % nv; if (b == a) b = (a + 1) % nv; if (c == a || c == b) c = (std::max(a, b) + 1) % nv; if (c == a || c == b) c = (c + 1) % nv;
            pr.idx.push_back(a); pr.idx.push_back(b); pr.idx.push_back(c);
            printf("xidx %u %u %u %u\n", p, a, b, c);
        }
        pr.emis.assign(nt, 0);
        static_assert(sizeof(bool) == 1, "emissive flags are one byte");
        FindEmissivesGpu(pr.v.data(), pr.idx.data(), reinterpret_cast<bool*>(pr.emis.data()), &mats[pr.mat], (uint32_t)pr.idx.size(), &pr.numLights);
        printf("xprim %u %u %u\n", p, pr.mat, pr.numLights);
        for (unsigned t = 0; t < nt; t++) printf("xemis %u %u %u\n", p, t, (unsigned)pr.emis[t]);
    }
    // ---- instances = scene data table entries (one primitive per mesh): DevicePrimitiveInstance (ModelStructs.h:73-80; PTMeshInstance.cpp:123-178)
    const unsigned NI = 12;
    std::vector<DevicePrimitiveInstance> table(NI);
    for (unsigned i = 0; i < NI; i++) {
        DevicePrimitiveInstance& e = table[i];
        const unsigned p = i % NP;
        e.m_Primitive = DevicePrimitive{prims[p].v.data(), prims[p].idx.data(), reinterpret_cast<bool*>(prims[p].emis.data()), &mats[prims[p].mat]};
        float m[16];
        { const float3 ax = unitvec(); const float ang = U() * 6.f; const float s = sinf(ang), c = cosf(ang), t = 1.f - c;
          const float sx = 0.5f + U(), sy = i % 2 ? sx : 0.5f + U(), sz = i % 2 ? sx : 0.5f + U();       // some non-uniform scales: normals go through the same matrix (quirk 20)
          const float r[9] = {t*ax.x*ax.x + c, t*ax.x*ax.y - s*ax.z, t*ax.x*ax.z + s*ax.y, t*ax.x*ax.y + s*ax.z, t*ax.y*ax.y + c, t*ax.y*ax.z - s*ax.x, t*ax.x*ax.z - s*ax.y, t*ax.y*ax.z + s*ax.x, t*ax.z*ax.z + c};
          const float tr[3] = {U()*6.f - 3.f, U()*6.f - 3.f, U()*6.f - 3.f};
          for (int a = 0; a < 3; a++) { m[4*a] = r[3*a] * sx; m[4*a+1] = r[3*a+1] * sy; m[4*a+2] = r[3*a+2] * sz; m[4*a+3] = tr[a]; }
          m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f; }
        e.m_Transform = sutil::Matrix4x4(m);
        const int mode = i == 9 ? 1 : (i == 4 || i == 7) ? 2 : 0;                          // ENABLED; the second instance of the emissive primitive DISABLED; two OVERRIDE
        e.m_EmissionMode = static_cast<Lumen::EmissionMode>(mode);
        const float rad[3] = {mode == 2 ? 1.f + 9.f * U() : 0.f, mode == 2 ? 1.f + 9.f * U() : 0.f, mode == 2 ? 1.f + 9.f * U() : 0.f}, scale = 0.5f + 2.f * U();
        e.m_EmissiveColorAndScale = make_float4(rad[0], rad[1], rad[2], scale);
        printf("xinst %u %u %d", i, p, mode); for (float v : m) printf(" %u", bits(v)); printf(" %u %u %u %u\n", bits(rad[0]), bits(rad[1]), bits(rad[2]), bits(scale));
    }
    SceneDataTableAccessor accessor((int)sizeof(DevicePrimitiveInstance), table.data());

    // ---- hit records + rays -> ExtractSurfaceDataGpu; set 0 = primary wave (origin = eye, contribution 1), set 1 = a deeper wave (random origins and contributions)
    const float3 eye = make_float3(0.3f, 1.1f, 4.2f);
    printf("xeye"); p3(eye); printf("\n");
    std::vector<SurfaceData> surf0;
    for (int set = 0; set < 2; set++) {
        auto* hits = makeAtomic<IntersectionData>(N); auto* rays = makeAtomic<IntersectionRayData>(N);
        hits->counter = N; rays->counter = N;
        std::vector<SurfaceData> out(N);
        memset(out.data(), 0, N * sizeof(SurfaceData));                                  // WaveFrontRenderer.cpp:652 / :818: the target buffer is zero-filled
        for (unsigned k = 0; k < N; k++) {
            const unsigned x = k % W, y = k / W;
            const unsigned entry = rng() % NI, prim = rng() % (unsigned)(prims[entry % NP].idx.size() / 3);
            float bu = U(), bv = U() * (1.f - bu);
            if (k % 17 == 3) { bu = 0.f; bv = 0.f; } if (k % 19 == 4) { bu = 1.f; bv = 0.f; }
            const float t = (k % 11 == 7) ? -1.f : 0.05f + 12.f * U();                    // misses carry t = -1 (WaveFrontShaders.cu:63-76)
            const __half hu = __float2half(bu), hv = __float2half(bv);
            __half2 bary; bary.x = hu; bary.y = hv;
            hits->data[k] = IntersectionData(k, t, bary, prim, entry, PixelIndex{(unsigned short)x, (unsigned short)y});
            const float3 dir = unitvec();
            const float3 org = set ? make_float3(U()*4.f - 2.f, U()*4.f - 2.f, U()*4.f - 2.f) : eye;
            const float3 con = set ? make_float3(U(), U(), U()) : make_float3(1.f, 1.f, 1.f);
            rays->data[k] = IntersectionRayData(PixelIndex{(unsigned short)x, (unsigned short)y}, org, dir, con);
            unsigned short ub, vb; memcpy(&ub, &hu, 2); memcpy(&vb, &hv, 2);
            printf("xhit %u %d %u %u %u %u %u %u %u", k, set, entry, prim, (unsigned)ub, (unsigned)vb, bits(t), x, y); p3(org); p3(dir); p3(con); printf("\n");
        }
        launch1d((N + 255u) / 256u, 256u, [&] { ExtractSurfaceDataGpu(N, hits, rays, out.data(), make_uint2(W, H), &accessor); });
        // The kernel builds its result in a local `SurfaceData output;` (no initialiser) and the two early exits store it with only some members set: an emitter seen
        // directly gets flags, pixel, t, normal and colour (:120-136), an alpha cut-out flags, position, incoming, transport, pixel, t, normal (:139-151).  The rest is
        // whatever the stack held (not reproducible, not read by any later kernel: both flags end the surface's shading); those cells are zeroed here, which is the
        // value this build's zero-initialised records hold (decision D5).
        for (unsigned k = 0; k < N; k++) {
            SurfaceData& z = out[k];
            const float3 zero = make_float3(0.f, 0.f, 0.f);
            if (z.m_SurfaceFlags & SURFACE_FLAG_EMISSIVE) {
                z.m_Position = zero; z.m_GeometricNormal = zero; z.m_Tangent = zero; z.m_IncomingRayDirection = zero; z.m_TransportFactor = zero;
                z.m_MaterialData.m_Tint = make_float4(0.f); z.m_MaterialData.m_Transmittance = make_float4(0.f); z.m_MaterialData.m_Parameters = make_uint4(0u);
            } else if (z.m_SurfaceFlags & SURFACE_FLAG_ALPHA_TRANSPARENT) {
                z.m_GeometricNormal = zero; z.m_Tangent = zero;
                z.m_MaterialData.m_Color = make_float4(0.f); z.m_MaterialData.m_Tint = make_float4(0.f); z.m_MaterialData.m_Transmittance = make_float4(0.f); z.m_MaterialData.m_Parameters = make_uint4(0u);
            }
        }
        for (unsigned k = 0; k < N; k++) {
            const SurfaceData& s = out[k];
            const MaterialData& md = s.m_MaterialData;
            printf("xsurf %u %d %u %u", k, set, (unsigned)s.m_SurfaceFlags, bits(s.m_IntersectionT));
            p3(s.m_Position); p3(s.m_Normal); p3(s.m_GeometricNormal); p3(s.m_Tangent); p3(s.m_IncomingRayDirection); p3(s.m_TransportFactor);
            p4(md.m_Color); p4(md.m_Tint); p4(md.m_Transmittance); printf(" %u %u %u\n", md.m_Parameters.x, md.m_Parameters.y, md.m_Parameters.z);
        }
        if (set == 0) surf0 = out;
        free(hits); free(rays);
    }
    // ---- GenerateMotionVector on the depth-0 surfaces (MotionVectors.cu:8-55; launch shape CPUShadingKernels.cu:27-54 is 2-D over the image)
    {
        float mm[16];
        { const float a = 1.7778f, th = 1.0f, zn = 0.5f, zf = 10000.f;                   // a projection times a rigid inverse-camera, as WaveFrontRenderer.cpp:763-776 builds it
          float proj[16] = {1.f / (a * th), 0, 0, 0, 0, 1.f / th, 0, 0, 0, 0, -(zf + zn) / (zf - zn), -(2.f * zf * zn) / (zf - zn), 0, 0, -1.f, 0};
          const float ang = 0.07f, c = cosf(ang), s = sinf(ang);
          float view[16] = {c, 0, -s, 0.1f, 0, 1, 0, -1.05f, s, 0, c, -4.1f, 0, 0, 0, 1};
          for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float acc = 0.f; for (int k = 0; k < 4; k++) acc += proj[4*i+k] * view[4*k+j]; mm[4*i+j] = acc; } }
        printf("xmvm"); for (float v : mm) printf(" %u", bits(v)); printf("\n");
        sutil::Matrix4x4 M(mm);
        g_mvImage.assign(N, ushort2{0, 0}); g_mvWidth = W;
        launch2d((W + 15u) / 16u, (H + 15u) / 16u, 16u, 16u, [&] { GenerateMotionVector(1, surf0.data(), make_uint2(W, H), &M); });
        for (unsigned k = 0; k < N; k++) printf("xmv %u %u %u\n", k, (unsigned)g_mvImage[k].x, (unsigned)g_mvImage[k].y);
    }
    // ---- ResolveDirectLightHits on the depth-0 surfaces (GPUShadeDirect.cu:11-40; launch shape CPUShadingKernels.cu:60-77: 2-D over the image): emitters seen directly
    // store their colour in the (cleared) DIRECT channel
    {
        g_directImage.assign(N, ushort4{0, 0, 0, 0}); g_mvWidth = W;
        launch2d((W + 15u) / 16u, (H + 15u) / 16u, 16u, 16u, [&] { ResolveDirectLightHits(surf0.data(), make_uint2(W, H), 0); });
        for (unsigned k = 0; k < N; k++) printf("xres %u %u %u %u %u\n", k, (unsigned)g_directImage[k].x, (unsigned)g_directImage[k].y, (unsigned)g_directImage[k].z, (unsigned)g_directImage[k].w);
    }
    // ---- the per-frame light list: LightDataBuffer::BuildLightDataBuffer's instance list (LightDataBuffer.cpp:37-125) and launch shape (CPUDataBufferKernels.cu:36-56), then the kernel
    {
        std::vector<LightInstanceData> lid;
        unsigned numEmissivePrims = 0, total = 0; float avg = 0.f;
        for (unsigned i = 0; i < NI; i++) {
            const int mode = (int)table[i].m_EmissionMode; const Prim& pr = prims[i % NP];
            const bool meshEmissive = pr.numLights > 0;                                    // ILumenMesh::GetEmissiveness: any primitive with lights (one primitive per mesh here)
            if (mode != 1 && ((mode == 0 && meshEmissive) || mode == 2)) {
                const unsigned numTriangles = (unsigned)(pr.idx.size() / 3);
                avg = ((avg * (float)numEmissivePrims) + (float)numTriangles) / (float)(numEmissivePrims + 1);
                numEmissivePrims++; total += pr.numLights;
                lid.push_back(LightInstanceData{i, numTriangles, pr.numLights});
            }
        }
        auto* lights = makeAtomic<TriangleLight>(4096);
        const unsigned gridW = (unsigned)std::ceil((float)lid.size() / 8.f), gridH = (unsigned)std::ceil((float)(uint32_t)std::roundf(avg) / 64.f);
        launch2d(gridW, gridH, 8u, 64u, [&] { BuildLightDataBufferGPU(lid.data(), (uint32_t)lid.size(), &accessor, lights); });
        printf("xnlights %u %u\n", lights->counter, total);
        for (unsigned k = 0; k < lights->counter; k++) { const TriangleLight& t = lights->data[k]; printf("xlight %u", k); p3(t.p0); p3(t.p1); p3(t.p2); p3(t.normal); p3(t.radiance); printf(" %u\n", bits(t.area)); }
    }
    return 0;
}
