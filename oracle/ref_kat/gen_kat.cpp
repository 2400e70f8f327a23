// Golden-vector generator: compiles the REFERENCE's own header-only device math host-side
// (RandomUtilities.cuh, MaterialStructs.h, disney.cuh + ggxmdf/frosted/bsdf_math) through shim.h
// and prints known-answer rows. Container-only: /root/reference never travels to the GPU box;
// only the numbers (tests/golden/ref_kat.npz, built by make_kat.py) are committed.
// This file contains no reference source text - it only #includes it from /root/reference.
#include "shim.h"
#include "CUDAKernels/RandomUtilities.cuh"
#include "Shaders/CppCommon/MaterialStructs.h"
#include "CUDAKernels/disney.cuh"
#include <random>

static std::mt19937 rng(20261002u);
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(rng); }
static float3 unitvec() {
    for (;;) { float3 v = make_float3(U()*2-1, U()*2-1, U()*2-1); float l = length(v); if (l > 0.1f && l <= 1.f) return v / l; }
}
struct MatIn { float c[4], tint[3], lum, trn[3], ior, p[11]; };
// p: metallic subsurface specular roughness spectint anisotropic sheen sheentint clearcoat clearcoatgloss transmission
static MaterialData build(const MatIn& in) {
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
static MatIn randmat(int kind) {
    MatIn in{};
    for (int i = 0; i < 3; i++) { in.c[i] = U(); in.tint[i] = U(); in.trn[i] = U() * 2.f; }
    in.c[3] = 1.f; in.lum = 0.25f + U();
    in.ior = (kind & 1) ? 1.f / (1.1f + U()) : 1.f;          // ExtractSurfaceData stores 1/ior
    for (int i = 0; i < 11; i++) in.p[i] = 0.f;
    in.p[3] = 0.02f + 0.98f * U();                           // roughness in (0,1]
    switch (kind % 6) {
    case 0: in.p[0] = 0.f; break;                                       // plain diffuse dielectric (Cornell-like)
    case 1: in.p[0] = U(); in.p[2] = U(); in.p[4] = U(); break;         // metal/specular mix
    case 2: in.p[0] = U(); in.p[2] = U(); in.p[6] = U(); in.p[7] = U(); in.p[1] = U(); break; // + sheen + subsurface
    case 3: in.p[0] = U(); in.p[2] = U(); in.p[8] = U(); in.p[9] = U(); break;               // + clearcoat
    case 4: in.p[0] = U()*0.5f; in.p[2] = U(); in.p[5] = U(); in.p[10] = 0.2f + 0.8f*U(); in.ior = 1.f/(1.1f+U()); break; // transmission + aniso
    case 5: for (int i = 0; i < 11; i++) if (i != 3) in.p[i] = U(); in.ior = 1.f/(1.1f+U()); break;   // everything
    }
    return in;
}
static void printmat(const MatIn& in, const MaterialData& m) {
    for (float v : in.c) printf(" %.9g", v); for (float v : in.tint) printf(" %.9g", v); printf(" %.9g", in.lum);
    for (float v : in.trn) printf(" %.9g", v); printf(" %.9g", in.ior); for (float v : in.p) printf(" %.9g", v);
    printf(" %u %u %u", m.m_Parameters.x, m.m_Parameters.y, m.m_Parameters.z);
}
int main() {
    // --- RNG rows: seed -> WangHash, then 4 RandomInt states + 4 RandomFloat values
    for (int i = 0; i < 256; i++) {
        unsigned seed = (i < 8) ? (unsigned)i : rng();
        unsigned h = WangHash(seed); unsigned s = h; printf("rng %u %u", seed, h);
        float f[4]; unsigned st[4];
        for (int k = 0; k < 4; k++) { f[k] = RandomFloat(s); st[k] = s; }
        for (int k = 0; k < 4; k++) printf(" %u", st[k]);
        for (int k = 0; k < 4; k++) printf(" %.9g", f[k]);
        printf("\n");
    }
    // --- material pack rows
    for (int i = 0; i < 256; i++) {
        MatIn in = randmat(5); MaterialData m = build(in);
        printf("pack"); printmat(in, m);
        printf(" %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", m.GetMetallic(), m.GetSubSurface(), m.GetSpecular(), m.GetRoughness(),
               m.GetSpecTint(), m.GetAnisotropic(), m.GetSheen(), m.GetSheenTint(), m.GetClearCoat(), m.GetClearCoatGloss(), m.GetTransmission());
    }
    // --- EvaluateBSDF rows
    for (int i = 0; i < 1500; i++) {
        MatIn in = randmat(i); MaterialData m = build(in);
        float3 N = unitvec(), T = unitvec();
        float3 wo = unitvec(); if (dot(wo, N) < 0.f && (i % 7)) wo = wo * -1.f;   // mostly front side, some back side
        float3 wi = unitvec(); if (dot(wi, N) < 0.f && (i % 5)) wi = wi * -1.f;
        float pdf = 0.f; float3 b = EvaluateBSDF(m, N, T, wo, wi, pdf);
        printf("eval"); printmat(in, m);
        printf(" %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g", N.x,N.y,N.z, T.x,T.y,T.z, wo.x,wo.y,wo.z, wi.x,wi.y,wi.z);
        printf(" %.9g %.9g %.9g %.9g\n", b.x, b.y, b.z, pdf);
    }
    // --- SampleBSDF rows (called exactly as GPUShadeIndirect.cu:89-103 does: iN == N, distance 1)
    for (int i = 0; i < 1500; i++) {
        MatIn in = randmat(i); MaterialData m = build(in);
        float3 N = unitvec(), T = unitvec();
        float3 wo = unitvec(); if (dot(wo, N) < 0.f && (i % 7)) wo = wo * -1.f;
        float r0 = U(), r1 = U(), r2 = U();
        float3 wi = make_float3(0.f); float pdf = 0.f; bool spec = false;
        float3 b = SampleBSDF(m, N, N, T, wo, 1.f, r0, r1, r2, wi, pdf, spec);
        printf("samp"); printmat(in, m);
        printf(" %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g", N.x,N.y,N.z, T.x,T.y,T.z, wo.x,wo.y,wo.z, r0, r1, r2);
        printf(" %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d\n", b.x, b.y, b.z, wi.x, wi.y, wi.z, pdf, spec ? 1 : 0);
    }
    return 0;
}
