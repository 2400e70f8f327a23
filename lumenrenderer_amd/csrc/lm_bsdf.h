// lm_bsdf.h — Disney principled BSDF of the shading kernels, organised for the loops it runs in.
//
// The light loops of this renderer evaluate ONE surface seen from ONE direction against MANY light directions: 32 candidates per
// pixel in the candidate pick, up to 5 in a spatial reuse pass, 2 in every reservoir merge.  The model is therefore split in two:
//
//   lm_lobes_setup(material, N, T, wo)  -> LmLobes    once per surface: shading frame, view direction in the frame, lobe mix,
//                                                     roughness -> alpha, the view-side masking terms (lambda(wo), G1(wo)),
//                                                     the view-side Fresnel / retro-reflection factors, tinted F0
//   lm_lobes_eval(LmLobes, wi)          -> f, pdf     per light direction: only what really depends on wi
//
// and sampling (path continuation) draws wi from one lobe and scores the other lobes with the same per-lobe functions.
//
// WHAT is computed follows the reference's LumenPT/src/CUDAKernels/{disney,ggxmdf,frosted,bsdf_math}.cuh (Lighthouse2 / appleseed
// lineage: Burley diffuse + subsurface blend, sheen, anisotropic GGX with Heitz visible-normal sampling, GTR1 clear coat, rough
// dielectric) and the 8-bit parameter packing of Shaders/CppCommon/MaterialStructs.h:84-217; tests/golden/ref_kat.npz holds
// vectors generated from those headers.  Locals the reference leaves uninitialised on early returns are zero here.
//
// Arithmetic policy.  Every function is a template over `A`:
//   LmExact  IEEE division / square root, every operation as specified by lm_math.h — the contract the bit-for-bit parity suite
//            checks against the oracle (default everywhere);
//   LmFast   v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp, no refinement sequences) — selectable for the ReSTIR target function
//            (tuning key "fast_resample"), where only a 1e-3 relative-L2 agreement is required (BASELINE north_star).  For the
//            isotropic opaque stack (diffuse + sheen + GGX specular, no clear coat) the fast policy goes further: LmQuick / lm_quick_eval
//            evaluate the same model in contracted form (no tangent frame: three cosines from N.wo, N.wi, wo.wi; |wo.h| cancelled out of the
//            pdf, G and 1 / cos merged into one reciprocal, FMA contraction): 2 rsq + 4 rcp + 1 sqrt per light direction.
// With LmExact the hoisting changes no operation and no operand: results equal the unsplit evaluation bit for bit.
#pragma once
#include "lm_math.h"

#define LM_PI      3.14159265358979323846264f
#define LM_INVPI   0.31830988618379067153777f
#define LM_TWOPI   6.28318530717958647692528f
#define LM_EPSILON 0.0001f      // the EPSILON macro seen by the reference's kernel bodies (bsdf_math.cuh:39-41)

struct LmExact {
    static constexpr bool contracted = false;
    static LM_HD float rcp(float x) { return 1.0f / x; }
    static LM_HD float div(float a, float b) { return a / b; }
    static LM_HD float sqrt(float x) { return sqrtf(x); }
    static LM_HD float rsqrt(float x) { return 1.0f / sqrtf(x); }
};
struct LmFast {
    static constexpr bool contracted = true;       // the light loops may use the algebraically contracted forms (LmQuick below)
#if defined(__HIP_DEVICE_COMPILE__)
    static LM_HD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static LM_HD float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static LM_HD float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static LM_HD float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else       // host builds (tests) have no such instructions; the fast policy is a device matter
    static LM_HD float rcp(float x) { return 1.0f / x; }
    static LM_HD float div(float a, float b) { return a / b; }
    static LM_HD float sqrt(float x) { return sqrtf(x); }
    static LM_HD float rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
};
template <class A> LM_HD lf3 lm_unit(const lf3& v) { return v * A::rsqrt(dot3(v, v)); }
template <class A> LM_HD lf3 lm_scale_inv(const lf3& v, float s) { return v * A::rcp(s); }      // v / s as sutil writes it: v * (1 / s)

// Shading-time material: the 80-byte MaterialData minus the emissive vector (never read while shading).
struct LmMaterial {
    float4 color;          // albedo, w alpha
    float4 transmittance;  // w: refractive-index slot (eta = 1/ior after surface extraction)
    float4 tint;           // w: luminance
    uint32_t p0, p1, p2;   // byte-packed parameters, words x/y/z of MaterialData::m_Parameters
};
// byte slots
#define LM_P_METALLIC(m)       lm_unpack8((m).p0, 0)
#define LM_P_SUBSURFACE(m)     lm_unpack8((m).p0, 8)
#define LM_P_SPECULAR(m)       lm_unpack8((m).p0, 16)
#define LM_P_ROUGHNESS(m)      lm_unpack8((m).p0, 24)
#define LM_P_SPECTINT(m)       lm_unpack8((m).p1, 0)
#define LM_P_ANISOTROPIC(m)    lm_unpack8((m).p1, 8)
#define LM_P_SHEEN(m)          lm_unpack8((m).p1, 16)
#define LM_P_SHEENTINT(m)      lm_unpack8((m).p1, 24)
#define LM_P_CLEARCOAT(m)      lm_unpack8((m).p2, 0)
#define LM_P_CLEARCOATGLOSS(m) lm_unpack8((m).p2, 8)
#define LM_P_TRANSMISSION(m)   lm_unpack8((m).p2, 16)
LM_HD float lm_unpack8(uint32_t w, uint32_t shift) { return (float)((w >> shift) & 255u) * (1.0f / 255.0f); }
LM_HD void lm_pack8(uint32_t& w, uint32_t shift, float v)
{
    const uint32_t q = (uint32_t)(v * 255.f);            // truncation (MaterialStructs.h:86)
    w &= ~(255u << shift);
    w |= q << shift;
}

// test hooks: 23 floats per material (color4 tint3 luminance transmittance3 ior + 11 parameters through the 8-bit setters)
LM_HD LmMaterial lm_material_from23(const float* m)
{
    LmMaterial sd;
    sd.color = make_float4(m[0], m[1], m[2], m[3]);
    sd.tint = make_float4(m[4], m[5], m[6], m[7]);
    sd.transmittance = make_float4(m[8], m[9], m[10], m[11]);
    sd.p0 = sd.p1 = sd.p2 = 0u;
    lm_pack8(sd.p0, 0, m[12]); lm_pack8(sd.p0, 8, m[13]); lm_pack8(sd.p0, 16, m[14]); lm_pack8(sd.p0, 24, m[15]);
    lm_pack8(sd.p1, 0, m[16]); lm_pack8(sd.p1, 8, m[17]); lm_pack8(sd.p1, 16, m[18]); lm_pack8(sd.p1, 24, m[19]);
    lm_pack8(sd.p2, 0, m[20]); lm_pack8(sd.p2, 8, m[21]); lm_pack8(sd.p2, 16, m[22]);
    return sd;
}

LM_HD float lm_schlick(float u) { const float m = saturatef(1.0f - u), m2 = sqrf(m), m4 = sqrf(m2); return m4 * m; }
LM_HD lf3 lm_to_frame(const lf3& V, const lf3& N, const lf3& T, const lf3& B) { return v3(dot3(V, T), dot3(V, B), dot3(V, N)); }
LM_HD lf3 lm_from_frame(const lf3& V, const lf3& N, const lf3& T, const lf3& B) { return V.x * T + V.y * B + V.z * N; }

// ---------------------------------------------------------------------------------------------------------------------
// The surface as the light loop sees it
// ---------------------------------------------------------------------------------------------------------------------
struct LmLobes {
    lf3 N, T, B;             // shading frame (T, B re-orthogonalised against N)
    lf3 wo, wol;             // view direction: world, frame
    float transmission;      // weight of the rough-dielectric lobe against the opaque stack
    float rough;             // perceptual roughness; <= 0.001 means the opaque stack is absent
    float w0, w1, w2, w3;    // lobe selection weights, normalised: diffuse, sheen, specular (GGX), clear coat (GTR1)
    // diffuse + sheen
    lf3 base;                // albedo
    float dielectric;        // 1 - metallic
    float subsurface;
    float cosO, fresO;       // N.wo and its Schlick weight
    lf3 sheenTint; float sheen;
    // specular, shared with the dielectric lobe: GGX alphas and the view-side masking
    float ax, ay, lamO, g1O;
    lf3 f0;                  // tinted normal-incidence reflectance (specular * 0.08 * tint, lerped to albedo by metallic)
    // clear coat: GTR1 with gamma = 1
    float coat;              // clearcoat parameter
    float cA2, cNorm, cLogA2, cLamO;      // alpha^2 (clamped), (alpha^2 - 1) / (pi ln alpha^2), ln alpha^2, lambda(wo)
    // dielectric lobe
    float ior, eta, rcpEta;  // eta as EVALUATION sees it: ior above the surface, 1 / ior below
};

// GGX (anisotropic): normal distribution and Smith lambda — ggxmdf.cuh:43-110
template <class A> LM_HD float lm_ggx_D(const lf3& m, float ax, float ay)
{
    if (m.z == 0) return sqrf(ax) * LM_INVPI;
    const float c2 = sqrf(m.z);
    const float st = A::sqrt(fmaxf(0.0f, 1 - c2));
    const float tan2 = A::div(1.0f - c2, c2);
    float stretched;
    if (ax == ay || st == 0.0f) stretched = A::rcp(sqrf(ax));
    else stretched = sqrf(A::div(m.x, st * ax)) + sqrf(A::div(m.y, st * ay));
    return A::rcp(LM_PI * ax * ay * sqrf(c2) * sqrf(1.0f + tan2 * stretched));
}
template <class A> LM_HD float lm_ggx_lambda(const lf3& v, float ax, float ay)
{
    if (v.z == 0) return 0;
    const float c2 = v.z * v.z;
    const float st = A::sqrt(fmaxf(0.0f, 1 - c2));
    float projected;
    if (ax == ay || st == 0.0f) projected = ax;
    else projected = A::sqrt(sqrf(A::div(v.x * ax, st)) + sqrf(A::div(v.y * ay, st)));
    const float tan2 = A::div(sqrf(st), c2);
    const float a2rcp = sqrf(projected) * tan2;
    return (-1.0f + A::sqrt(1.0f + a2rcp)) * 0.5f;
}
// GTR1 — ggxmdf.cuh:150-228.  The part that depends only on alpha is in LmLobes.
template <class A> LM_HD float lm_gtr1_lambda(const lf3& v, float a2, float logA2)
{
    if (v.z == 0) return 0;
    const float c2 = sqrf(v.z);
    const float st = A::sqrt(fmaxf(0.0f, 1.0f - c2));
    if (st == 0) return 0;
    const float cot2 = A::div(c2, sqrf(st));
    const float cot = A::sqrt(cot2);
    const float a = A::sqrt(cot2 + a2);
    const float b = A::sqrt(cot2 + 1.0f);
    const float c = lm_logf(cot + b);
    const float d = lm_logf(cot + a);
    return A::div(a - b + cot * (c - d), cot * logA2);
}
LM_HD void lm_alpha_from_roughness(float roughness, float anisotropy, float& ax, float& ay)      // ggxmdf.cuh / disney.cuh:60-68
{
    const float sq = roughness * roughness;
    const float aspect = sqrtf(1.0f + anisotropy * (anisotropy < 0 ? 0.9f : -0.9f));
    ax = fmaxf(0.001f, sq / aspect);
    ay = fmaxf(0.001f, sq * aspect);
}

// Once per surface and view direction.  `N` is the shading normal the caller wants the frame built on (sampling passes the
// normal flipped to the side of wo, evaluation does not: disney.cuh:181-183 vs :320-340).
template <class A> LM_HD void lm_lobes_setup(const LmMaterial& sd, const lf3& N, const lf3& iT, const lf3& wow, LmLobes& L)
{
    L.N = N;
    L.B = lm_unit<A>(cross3(N, iT));
    L.T = lm_unit<A>(cross3(N, L.B));
    L.wo = wow;
    L.wol = lm_to_frame(wow, N, L.T, L.B);
    L.transmission = LM_P_TRANSMISSION(sd);
    L.rough = LM_P_ROUGHNESS(sd);
    const float metallic = LM_P_METALLIC(sd);
    // lobe mix (disney.cuh:152-171): luminance, sheen, specular -> 1 with metallic, a quarter of the clear coat
    L.coat = LM_P_CLEARCOAT(sd);
    L.sheen = LM_P_SHEEN(sd);
    const float specular = LM_P_SPECULAR(sd);
    L.w0 = lerpf(sd.tint.w, 0.f, metallic);
    L.w1 = lerpf(L.sheen, 0.f, metallic);
    L.w2 = lerpf(specular, 1.f, metallic);
    L.w3 = L.coat * 0.25f;
    const float inv = A::rcp(L.w0 + L.w1 + L.w2 + L.w3);
    L.w0 *= inv; L.w1 *= inv; L.w2 *= inv; L.w3 *= inv;
    // diffuse / sheen
    L.base = v3(sd.color);
    L.dielectric = 1.0f - metallic;
    L.subsurface = LM_P_SUBSURFACE(sd);
    L.cosO = dot3(N, wow);
    L.fresO = lm_schlick(L.cosO);
    const float sheenTint = LM_P_SHEENTINT(sd);
    L.sheenTint = (1.0f - sheenTint) + sheenTint * v3(sd.tint);
    // specular
    lm_alpha_from_roughness(L.rough, LM_P_ANISOTROPIC(sd), L.ax, L.ay);
    L.lamO = lm_ggx_lambda<A>(L.wol, L.ax, L.ay);
    L.g1O = A::rcp(1.0f + L.lamO);
    const float specTint = LM_P_SPECTINT(sd);
    lf3 f0 = (1.0f - specTint) + specTint * v3(sd.tint);
    f0 = f0 * (specular * 0.08f);
    L.f0 = (1.0f - metallic) * f0 + metallic * v3(sd.color);
    // clear coat
    L.cA2 = 0.f; L.cNorm = 0.f; L.cLogA2 = 0.f; L.cLamO = 0.f;
    if (L.w3 > 0) {
        const float alpha = clampf(lerpf(0.1f, 0.001f, LM_P_CLEARCOATGLOSS(sd)), 0.001f, 0.999f);
        L.cA2 = sqrf(alpha);
        L.cLogA2 = lm_logf(L.cA2);
        L.cNorm = A::div(L.cA2 - 1.0f, LM_PI * L.cLogA2);
        L.cLamO = lm_gtr1_lambda<A>(L.wol, L.cA2, L.cLogA2);
    }
    // dielectric
    L.ior = sd.transmittance.w;
    L.eta = L.wol.z > 0 ? L.ior : A::rcp(L.ior);
    L.rcpEta = A::rcp(L.eta);
}

// What one light direction adds: the direction in the frame and the two half vectors (the world-space one feeds the diffuse and
// sheen lobes, the frame-space one the microfacet lobes; the reference normalises them separately and so do we)
struct LmDir { lf3 w, l, hw, hl; };

// ---- lobes: value and solid-angle pdf for one direction ---------------------------------------------------------------
// Burley diffuse with the Hanrahan-Krueger subsurface blend — disney.cuh:33-75
template <class A> LM_HD float lm_lobe_diffuse(const LmLobes& L, const lf3& wiw, const lf3& hw, lf3& value)
{
    const float cin = dot3(L.N, wiw), cih = dot3(wiw, hw);
    const float fl = lm_schlick(cin), fv = L.fresO;
    float fd = 0;
    if (L.subsurface != 1.0f) {
        const float fd90 = 0.5f + 2.0f * sqrf(cih) * L.rough;
        fd = lerpf(1.f, fd90, fl) * lerpf(1.f, fd90, fv);
    }
    if (L.subsurface > 0) {
        const float fss90 = sqrf(cih) * L.rough;
        const float fss = lerpf(1.0f, fss90, fl) * lerpf(1.0f, fss90, fv);
        const float ss = 1.25f * (fss * (A::rcp(fabsf(L.cosO) + fabsf(cin)) - 0.5f) + 0.5f);
        fd = lerpf(fd, ss, L.subsurface);
    }
    value = L.base * fd * LM_INVPI * L.dielectric;
    return fabsf(cin) * LM_INVPI;
}
// sheen — disney.cuh:77-92
LM_HD float lm_lobe_sheen(const LmLobes& L, const lf3& wiw, const lf3& hw, lf3& value)
{
    const float fh = lm_schlick(dot3(wiw, hw));
    value = L.sheenTint * (fh * L.sheen * L.dielectric);
    return 1.0f / (2 * LM_PI);
}
// GGX specular — disney.cuh:94-150 (evaluate branch), microfacet pdf through the visible-normal distribution
template <class A> LM_HD float lm_lobe_specular(const LmLobes& L, const lf3& wil, const lf3& m, lf3& value)
{
    if (L.wol.z == 0 || wil.z == 0) return 0;
    const float coh = dot3(L.wol, m);
    if (coh == 0) return 0;
    const float D = lm_ggx_D<A>(m, L.ax, L.ay);
    const float G = A::rcp(1.0f + L.lamO + lm_ggx_lambda<A>(wil, L.ax, L.ay));
    const float fh = lm_schlick(fabsf(coh));
    value = ((1.0f - fh) * L.f0 + fh) * A::div(D * G, fabsf(4.0f * L.wol.z * wil.z));
    return A::div(A::div(L.g1O * fabsf(coh) * D, fabsf(L.wol.z)), fabsf(4.0f * coh));
}
// GTR1 clear coat
template <class A> LM_HD float lm_coat_D(const LmLobes& L, const lf3& m) { return L.cNorm * A::rcp(1 + (L.cA2 - 1) * sqrf(m.z)); }
template <class A> LM_HD float lm_lobe_coat(const LmLobes& L, const lf3& wil, const lf3& m, lf3& value)
{
    if (L.wol.z == 0 || wil.z == 0) return 0;
    const float coh = dot3(L.wol, m);
    if (coh == 0) return 0;
    const float D = lm_coat_D<A>(L, m);
    const float G = A::rcp(1.0f + L.cLamO + lm_gtr1_lambda<A>(wil, L.cA2, L.cLogA2));
    value = v3(lerpf(0.04f, 1.0f, lm_schlick(fabsf(coh))) * 0.25f * L.coat) * A::div(D * G, fabsf(4.0f * L.wol.z * wil.z));
    return A::div(D * fabsf(m.z), fabsf(4.0f * coh));
}

// rough dielectric — frosted.cuh:28-120
LM_HD float lm_fresnel_dielectric(float eta, float ci, float ct)
{
    if (ci == 0 && ct == 0) return 1;
    const float k0 = eta * ct, k1 = eta * ci;
    return 0.5f * (sqrf((ci - k0) / (ci + k0)) + sqrf((ct - k1) / (ct + k1)));
}
LM_HD float lm_fresnel_reflectance(float ci, float eta, float& ct)
{
    const float st2 = (1 - sqrf(ci)) * sqrf(eta);
    if (st2 > 1) { ct = 0; return 1; }
    ct = fminf(sqrtf(fmaxf(1 - st2, 0.0f)), 1.0f);
    return lm_fresnel_dielectric(eta, fabsf(ci), ct);
}
LM_HD float lm_ggx_vndf_pdf(const LmLobes& L, const lf3& m)      // pdf of the visible-normal distribution around wo
{
    if (L.wol.z == 0.0f) return 0;
    return L.g1O * fabsf(dot3(L.wol, m)) * lm_ggx_D<LmExact>(m, L.ax, L.ay) / fabsf(L.wol.z);
}
LM_HD lf3 lm_glass_reflect(const LmLobes& L, const lf3& wil, const lf3& m, float F)
{
    const float denom = fabsf(4 * L.wol.z * wil.z);
    if (denom == 0) return v3(0);
    const float D = lm_ggx_D<LmExact>(m, L.ax, L.ay), G = 1.0f / (1.0f + L.lamO + lm_ggx_lambda<LmExact>(wil, L.ax, L.ay));
    return L.base * (F * D * G / denom);
}
LM_HD lf3 lm_glass_refract(const LmLobes& L, float eta, const lf3& wil, const lf3& m, float T)
{
    if (L.wol.z == 0 || wil.z == 0) return v3(0);
    const float cih = dot3(m, wil), coh = dot3(m, L.wol);
    const float dots = (cih * coh) / (wil.z * L.wol.z);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return v3(0);
    const float D = lm_ggx_D<LmExact>(m, L.ax, L.ay), G = 1.0f / (1.0f + L.lamO + lm_ggx_lambda<LmExact>(wil, L.ax, L.ay));
    float mult = fabsf(dots) * T * D * G / sqrf(sd);
    mult *= sqrf(eta);                                   // radiance transport (the reference never passes adjoint = true)
    return L.base * mult;
}
LM_HD float lm_refraction_jacobian(const LmLobes& L, const lf3& wil, const lf3& m, float eta)
{
    const float cih = dot3(m, wil), coh = dot3(m, L.wol);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return 0;
    return fabsf(cih) * sqrf(eta / sd);
}
LM_HD lf3 lm_upper(const lf3& h) { return h.z < 0 ? (h * -1.f) : h; }
// value and pdf of the dielectric lobe for a direction on either side (disney.cuh:336-372).  Rare on this path (no benchmark
// material transmits), so it stays on the exact policy.
LM_HD float lm_lobe_glass(const LmLobes& L, const lf3& wil, lf3& value)
{
    lf3 m; float pdf, jacobian, ct;
    if (wil.z * L.wol.z >= 0) {
        m = lm_upper(normalize3(wil + L.wol));
        const float com = dot3(L.wol, m);
        const float F = lm_fresnel_reflectance(com, L.rcpEta, ct);
        value = lm_glass_reflect(L, wil, m, F);
        pdf = F;                                         // reflection probability F / (F + (1 - F)), written as the reference computes it:
        { const float r = F * 1.f, t = (1 - F) * 1.f, sum = r + t; pdf = sum != 0 ? r / sum : 1; }
        jacobian = com == 0 ? 0 : 1 / (4 * fabsf(com));
    } else {
        m = lm_upper(normalize3(L.wol + L.eta * wil));
        const float com = dot3(L.wol, m);
        const float F = lm_fresnel_reflectance(com, L.rcpEta, ct);
        value = lm_glass_refract(L, L.eta, wil, m, 1 - F);
        { const float r = F * 1.f, t = (1 - F) * 1.f, sum = r + t; pdf = 1 - (sum != 0 ? r / sum : 1); }
        jacobian = lm_refraction_jacobian(L, wil, m, L.eta);
    }
    return pdf * (jacobian * lm_ggx_vndf_pdf(L, m));
}

// ---------------------------------------------------------------------------------------------------------------------
// evaluation for one light direction — what disney.cuh:320-405 returns for (material, N, T, wo, wi)
// ---------------------------------------------------------------------------------------------------------------------
template <class A> LM_HD lf3 lm_lobes_eval(const LmLobes& L, const lf3& wiw, float& pdf)
{
    lf3 glass = v3(0);
    float glassPdf = 0.f;
    if (L.transmission > 0.f) {
        if (L.eta == 1) { pdf = 0; return v3(0); }
        glassPdf = lm_lobe_glass(L, lm_to_frame(wiw, L.N, L.T, L.B), glass);
    }
    if (L.rough <= 0.001f) { pdf = glassPdf; return glass; }
    pdf = 0;
    lf3 value = v3(0);
    if (L.w0 + L.w1 > 0) {
        const lf3 hw = lm_unit<A>(wiw + L.wo);
        if (L.w0 > 0) pdf += L.w0 * lm_lobe_diffuse<A>(L, wiw, hw, value);
        if (L.w1 > 0) pdf += L.w1 * lm_lobe_sheen(L, wiw, hw, value);       // replaces the diffuse value: reference behaviour (disney.cuh:373)
    }
    if (L.w2 + L.w3 > 0) {
        const lf3 wil = lm_to_frame(wiw, L.N, L.T, L.B);
        const lf3 hl = lm_unit<A>(L.wol + wil);
        if (L.w2 > 0) {
            lf3 c = v3(0);
            const float p = lm_lobe_specular<A>(L, wil, hl, c);
            if (p > 0) { pdf += L.w2 * p; value = value + c; }
        }
        if (L.w3 > 0) {
            lf3 c = v3(0);
            const float p = lm_lobe_coat<A>(L, wil, hl, c);
            if (p > 0) { pdf += L.w3 * p; value = value + c; }
        }
    }
    pdf = (pdf * (1.f - L.transmission));
    pdf += (glassPdf * L.transmission);
    return (glass * L.transmission) + (value * (1.f - L.transmission));
}
// one-shot form (next-event estimation evaluates a surface for a single light direction)
template <class A = LmExact> LM_HD lf3 lm_evaluate_bsdf(const LmMaterial& sd, const lf3& iN, const lf3& iT, const lf3& wow, const lf3& wiw, float& pdf)
{
    LmLobes L;
    lm_lobes_setup<A>(sd, iN, iT, wow, L);
    return lm_lobes_eval<A>(L, wiw, pdf);
}

// ---------------------------------------------------------------------------------------------------------------------
// Contracted evaluation (fast policy only) of the ISOTROPIC opaque stack without clear coat.  Same model, different algebra — no tangent
// frame at all, because with alpha_x = alpha_y every term depends on three cosines only (N.wi, N.h, wi.h):
//   * h = (wo + wi) / |wo + wi| in world space;   wi.h = wo.h;   N.h = (N.wo + N.wi) / |wo + wi|
//   * D  = 1 / (pi a^2 ((1 - (N.h)^2) / a^2 + (N.h)^2)^2)                                one reciprocal, no tan / sin
//   * lambda(wi) = (sqrt(1 + a^2 (1 - (N.wi)^2) / (N.wi)^2) - 1) / 2                     one reciprocal, one square root
//   * G / cos_i  = 1 / ((1 + lambda(wo) + lambda(wi)) cos_i)                             one reciprocal
//   * specular pdf = G1(wo) |wo.h| D / |cos_o| / (4 |wo.h|) = D * [G1(wo) / (4 |cos_o|)]    no reciprocal: the bracket is per surface
// Agreement with the exact policy: a few ulp (tests/test_gpu_parity.py holds whole frames to 1e-3 relative L2; measured 1e-8).
// ---------------------------------------------------------------------------------------------------------------------
struct LmQuick {
    lf3 N, wo;                       // shading normal, view direction (world)
    float cosO, fresO;               // N.wo and its Schlick weight
    float w0, w1, w2;                // lobe weights: diffuse, sheen, specular (normalised; no clear coat on this path)
    float rough, subsurface;
    lf3 kd;                          // albedo (1 - metallic) / pi
    lf3 sheenTint, f0;
    float pdfDiffuse, pdfSheen;      // w0 / pi,  w1 / (2 pi)
    float sheenK;                    // sheen (1 - metallic)
    float a2, ia2, kD;               // alpha^2, 1 / alpha^2, 1 / (pi alpha^2)
    float kG;                        // 1 + lambda(wo)
    float inv4cosO, pdfSpecular;     // 1 / (4 |cos_o|),  w2 G1(wo) / (4 |cos_o|); both 0 for an exactly grazing view (no specular term, as in the exact path)
};
// Does the contracted evaluation cover this material?  Not with a dielectric lobe, a clear coat, anisotropy, or a roughness byte of 0
// (the opaque stack is then absent, disney.cuh:374): such surfaces are scored by the exact path (lm_restir.h LM_RARE).
LM_HD bool lm_quick_contracts(const LmMaterial& m) { return (m.p2 & 0x00ff00ffu) == 0u && (m.p0 >> 24) != 0u && (m.p1 & 0x0000ff00u) == 0u; }
// Once per surface, from the material, the shading normal and the view direction only: neither the tangent nor the dielectric constants
// of the surface record are touched (a reuse pass does not even load them).
LM_HD void lm_quick_setup(const LmMaterial& sd, const lf3& N, const lf3& wow, LmQuick& Q)
{
#pragma clang fp contract(fast)
    Q.N = N; Q.wo = wow;
    Q.cosO = dot3(N, wow);
    Q.fresO = lm_schlick(Q.cosO);
    Q.rough = LM_P_ROUGHNESS(sd);
    Q.subsurface = LM_P_SUBSURFACE(sd);
    const float metallic = LM_P_METALLIC(sd), dielectric = 1.0f - metallic, sheen = LM_P_SHEEN(sd), specular = LM_P_SPECULAR(sd);
    Q.w0 = sd.tint.w * dielectric; Q.w1 = sheen * dielectric; Q.w2 = specular + metallic * (1.0f - specular);
    const float inv = LmFast::rcp(Q.w0 + Q.w1 + Q.w2);
    Q.w0 *= inv; Q.w1 *= inv; Q.w2 *= inv;
    Q.kd = v3(sd.color) * (LM_INVPI * dielectric);
    const float sheenTint = LM_P_SHEENTINT(sd), specTint = LM_P_SPECTINT(sd);
    Q.sheenTint = (1.0f - sheenTint) + sheenTint * v3(sd.tint);
    Q.f0 = dielectric * (((1.0f - specTint) + specTint * v3(sd.tint)) * (specular * 0.08f)) + metallic * v3(sd.color);
    Q.pdfDiffuse = Q.w0 * LM_INVPI;
    Q.pdfSheen = Q.w1 * (0.5f * LM_INVPI);
    Q.sheenK = sheen * dielectric;
    Q.a2 = sqrf(fmaxf(0.001f, Q.rough * Q.rough));      // alpha_x = alpha_y = max(0.001, roughness^2) without anisotropy
    Q.ia2 = LmFast::rcp(Q.a2);
    Q.kD = Q.ia2 * LM_INVPI;
    const float c2 = Q.cosO * Q.cosO;
    const float lamO = Q.cosO != 0.f ? 0.5f * (LmFast::sqrt(1.0f + Q.a2 * fmaxf(0.0f, 1.0f - c2) * LmFast::rcp(c2)) - 1.0f) : 0.f;
    Q.kG = 1.0f + lamO;
    Q.inv4cosO = Q.cosO != 0.f ? LmFast::rcp(4.0f * fabsf(Q.cosO)) : 0.f;
    Q.pdfSpecular = Q.w2 * LmFast::rcp(Q.kG) * Q.inv4cosO;
}
// `cin` = N.wi > 0 (the caller has culled lights below the horizon)
LM_HD lf3 lm_quick_eval(const LmQuick& Q, const lf3& wiw, float cin, float& pdf)
{
#pragma clang fp contract(fast)
    const lf3 h = Q.wo + wiw;                            // (from the components, not from 2 + 2 wo.wi: that loses every digit when wo ~ -wi)
    const float hinv = LmFast::rsqrt(dot3(h, h));
    const float ch = dot3(wiw, h) * hinv;                // wi.h = wo.h
    const float hz = (Q.cosO + cin) * hinv;              // N.h
    lf3 value = v3(0.f);
    pdf = 0.f;
    if (Q.w0 + Q.w1 > 0.f) {
        if (Q.w0 > 0.f) {
            const float fl = lm_schlick(cin), ch2r = ch * ch * Q.rough;
            float fd = 0.f;
            if (Q.subsurface != 1.0f) { const float k = 2.0f * ch2r - 0.5f; fd = (1.0f + k * fl) * (1.0f + k * Q.fresO); }
            if (Q.subsurface > 0.f) {
                const float k = ch2r - 1.0f, fss = (1.0f + k * fl) * (1.0f + k * Q.fresO);
                const float ss = 1.25f * (fss * (LmFast::rcp(fabsf(Q.cosO) + cin) - 0.5f) + 0.5f);
                fd = fd + Q.subsurface * (ss - fd);
            }
            value = Q.kd * fd;
            pdf = Q.pdfDiffuse * cin;
        }
        if (Q.w1 > 0.f) { value = Q.sheenTint * (lm_schlick(ch) * Q.sheenK); pdf += Q.pdfSheen; }      // replaces the diffuse value, as the exact path does
    }
    if (Q.w2 > 0.f && ch != 0.f) {
        // sin^2 of the half vector against N from its tangential part h - (h.N) N, NOT as 1 - (N.h)^2: near the specular peak of a smooth
        // surface (alpha^2 ~ 1e-4) the latter loses every digit the division by alpha^2 then magnifies (4e-4 relative error in D against the
        // reference's tangent-frame form on tests/golden/ref_kat.npz; 1e-6 this way, for six more operations per light point)
        const float hn = Q.cosO + cin;                   // h.N, unnormalised
        const lf3 ht = h - Q.N * hn;
        const float hz2 = hz * hz, e = dot3(ht, ht) * (hinv * hinv) * Q.ia2 + hz2;
        const float D = Q.kD * LmFast::rcp(e * e);
        const float c2 = cin * cin;
        const float lamI = 0.5f * (LmFast::sqrt(1.0f + Q.a2 * (1.0f - c2) * LmFast::rcp(c2)) - 1.0f);
        const float gOverCos = LmFast::rcp((Q.kG + lamI) * cin);
        const float fh = lm_schlick(fabsf(ch));
        const float p = Q.pdfSpecular * D;
        if (p > 0.f) { pdf += p; value = value + ((1.0f - fh) * Q.f0 + fh) * (D * gOverCos * Q.inv4cosO); }
    }
    return value;
}

// ---------------------------------------------------------------------------------------------------------------------
// sampling (path continuation) — what disney.cuh:173-304 returns for (material, iN, N, T, wo, distance, r0, r1, r2)
// ---------------------------------------------------------------------------------------------------------------------
// Heitz visible-normal sampling of the GGX distribution (ggxmdf.cuh:112-148)
LM_HD lf3 lm_ggx_sample(const lf3& v, float r0, float r1, float ax, float ay)
{
    const float sgn = v.z < 0.0f ? -1.0f : 1.0f;
    const lf3 stretched = normalize3(v3(sgn * v.x * ax, sgn * v.y * ay, sgn * v.z));
    const lf3 t1 = v.z < 0.9999f ? normalize3(cross3(stretched, v3(0, 0, 1))) : v3(1, 0, 0);
    const lf3 t2 = cross3(t1, stretched);
    const float a = 1.0f / (1.0f + stretched.z);
    const float r = sqrtf(r0);
    const float phi = r1 < a ? (r1 / a * LM_PI) : (LM_PI + (r1 - a) / (1.0f - a) * LM_PI);
    float p1, p2;
    lm_sincosf(phi, &p2, &p1);
    p1 *= r;
    p2 *= r * (r1 < a ? 1.0f : stretched.z);
    const lf3 h = p1 * t1 + p2 * t2 + sqrtf(fmaxf(0.0f, 1.0f - p1 * p1 - p2 * p2)) * stretched;
    return normalize3(v3(h.x * ax, h.y * ay, fmaxf(0.0f, h.z)));
}
LM_HD lf3 lm_coat_sample(const LmLobes& L, float r0, float r1)
{
    const float c2 = (1.0f - lm_powf(L.cA2, 1.0f - r0)) / (1.0f - L.cA2);
    const float st = sqrtf(fmaxf(0.0f, 1.0f - c2));
    float cphi, sphi;
    lm_sincosf(LM_TWOPI * r1, &sphi, &cphi);
    return v3(cphi * st, sphi * st, sqrtf(c2));
}
// draw wi by mirroring wo about a sampled micro-normal; value = Fresnel * D * G (the caller divides by |4 cos cos|)
template <bool GGX> LM_HD void lm_draw_microfacet(const LmLobes& L, float r0, float r1, lf3& wil, float& pdf, lf3& value)
{
    if (L.wol.z == 0) { value = v3(0); pdf = 0; return; }
    const lf3 m = GGX ? lm_ggx_sample(L.wol, r0, r1, L.ax, L.ay) : lm_coat_sample(L, r0, r1);
    wil = reflect3(L.wol * -1.0f, m);
    if (wil.z == 0) return;
    const float coh = dot3(L.wol, m);
    const float D = GGX ? lm_ggx_D<LmExact>(m, L.ax, L.ay) : lm_coat_D<LmExact>(L, m);
    pdf = (GGX ? L.g1O * fabsf(coh) * D / fabsf(L.wol.z) : D * fabsf(m.z)) / fabsf(4.0f * coh);
    if (pdf < 1.0e-6f) return;
    const float G = GGX ? 1.0f / (1.0f + L.lamO + lm_ggx_lambda<LmExact>(wil, L.ax, L.ay)) : 1.0f / (1.0f + L.cLamO + lm_gtr1_lambda<LmExact>(wil, L.cA2, L.cLogA2));
    const float fh = lm_schlick(fabsf(coh));
    value = GGX ? (1.0f - fh) * L.f0 + fh : v3(lerpf(0.04f, 1.0f, fh) * 0.25f * L.coat);
    value = value * (D * G);
}

LM_HD lf3 lm_sample_bsdf(const LmMaterial& sd, lf3 iN, const lf3& N, const lf3& iT, const lf3& wow, float distance,
                         float r0, float r1, float r2, lf3& wiw, float& pdf, bool& specular)
{
    const float flip = (dot3(wow, N) < 0) ? -1.f : 1.f;
    LmLobes L;
    lm_lobes_setup<LmExact>(sd, iN * flip, iT, wow, L);
    if (r0 < L.transmission) {
        // rough dielectric: micro-normal from the visible-normal distribution, then reflect or refract by the Fresnel term
        specular = true;
        const float r3 = r0 / L.transmission;
        const float eta = flip < 0 ? (1 / L.ior) : L.ior;
        if (eta == 1) return v3(0);
        const lf3 beer = v3(lm_expf(-sd.transmittance.x * distance * 2.0f),
                            lm_expf(-sd.transmittance.y * distance * 2.0f),
                            lm_expf(-sd.transmittance.z * distance * 2.0f));
        const lf3 m = lm_ggx_sample(L.wol, r1, r3, L.ax, L.ay);
        const float rcp_eta = 1 / eta, com = clampf(dot3(L.wol, m), -1.0f, 1.0f);
        float ct, jacobian;
        const float F = lm_fresnel_reflectance(com, eta, ct);
        lf3 wil, ret;
        if (r2 < F) {
            wil = reflect3(L.wol * -1.0f, m);
            if (wil.z * L.wol.z <= 0) return v3(0);
            ret = lm_glass_reflect(L, wil, m, F);
            pdf = F; jacobian = com == 0 ? 0 : 1 / (4 * fabsf(com));
        } else {
            // refracted direction with eta where 1 / eta is expected: reference behaviour (disney.cuh:219)
            wil = com > 0 ? (eta * com - ct) * m - eta * L.wol : (eta * com + ct) * m - eta * L.wol;
            wil = wil * ((3 - dot3(wil, wil)) * 0.5f);
            if (wil.z * L.wol.z > 0) return v3(0);
            ret = lm_glass_refract(L, rcp_eta, wil, m, 1 - F);
            pdf = 1 - F; jacobian = lm_refraction_jacobian(L, wil, m, rcp_eta);
        }
        pdf *= jacobian * lm_ggx_vndf_pdf(L, m);
        if (pdf > 1.0e-6f) wiw = lm_from_frame(wil, L.N, L.T, L.B);
        return ret * beer;
    }
    // opaque stack: one lobe proposes the direction, every lobe scores it
    const float r3 = (r0 - L.transmission) / (1 - L.transmission);
    const float cdfx = L.w0, cdfy = L.w0 + L.w1, cdfz = L.w0 + L.w1 + L.w2;
    float probability = 0.f;
    lf3 value = v3(0);
    int drawn;                                           // lobe that proposed the direction: 0 diffuse 1 sheen 2 specular 3 coat
    if (r3 < cdfy) {
        const float rr = r3 / cdfy;
        const float term1 = LM_TWOPI * rr, term2 = sqrtf(1 - r1);
        float s, c;
        lm_sincosf(term1, &s, &c);
        wiw = (c * term2 * L.T) + (s * term2) * L.B + sqrtf(r1) * L.N;        // cosine-weighted hemisphere
        const lf3 hw = normalize3(wiw + wow);
        if (r3 < cdfx) { drawn = 0; probability = L.w0 * lm_lobe_diffuse<LmExact>(L, wiw, hw, value); }
        else { drawn = 1; probability = L.w1 * lm_lobe_sheen(L, wiw, hw, value); }
    } else {
        lf3 wil = v3(0);
        float lobePdf = 0.f;
        if (r3 < cdfz) { drawn = 2; lm_draw_microfacet<true>(L, (r3 - cdfy) / (cdfz - cdfy), r1, wil, lobePdf, value); probability = L.w2 * lobePdf; }
        else { drawn = 3; lm_draw_microfacet<false>(L, (r3 - cdfz) / (1 - cdfz), r1, wil, lobePdf, value); probability = L.w3 * lobePdf; }
        value = value * (1.0f / fabsf(4.0f * L.wol.z * wil.z));
        wiw = lm_from_frame(wil, L.N, L.T, L.B);
    }
    const float w0 = drawn == 0 ? 0.f : L.w0, w1 = drawn == 1 ? 0.f : L.w1, w2 = drawn == 2 ? 0.f : L.w2, w3 = drawn == 3 ? 0.f : L.w3;
    if (w0 + w1 > 0) {
        const lf3 hw = normalize3(wiw + wow);
        lf3 c;
        if (w0 > 0) { c = v3(0); probability += w0 * lm_lobe_diffuse<LmExact>(L, wiw, hw, c); value = value + c; }
        if (w1 > 0) { c = v3(0); probability += w1 * lm_lobe_sheen(L, wiw, hw, c); value = value + c; }
    }
    if (w2 + w3 > 0) {
        const lf3 wil = lm_to_frame(wiw, L.N, L.T, L.B);
        const lf3 hl = normalize3(L.wol + wil);
        lf3 c;
        if (w2 > 0) { c = v3(0); probability += w2 * lm_lobe_specular<LmExact>(L, wil, hl, c); value = value + c; }
        if (w3 > 0) { c = v3(0); probability += w3 * lm_lobe_coat<LmExact>(L, wil, hl, c); value = value + c; }
    }
    if (probability > 1.0e-6f) pdf = probability; else pdf = 0;
    return value;
}
