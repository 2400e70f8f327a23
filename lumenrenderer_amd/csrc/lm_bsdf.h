// lm_bsdf.h — Disney principled BSDF for the shading kernels.
//
// Behaviour follows the reference's LumenPT/src/CUDAKernels/{disney,ggxmdf,frosted,bsdf_math}.cuh and the 8-bit
// parameter packing of LumenPT/src/Shaders/CppCommon/MaterialStructs.h:84-217 (the CUDA build takes the
// __CUDACC__ sincosf branches; so does this).  Locals the reference leaves uninitialised on early returns
// (disney.cuh:97,104,117,119) are zero here.  All arithmetic obeys lm_math.h.
#pragma once
#include "lm_math.h"

#define LM_PI      3.14159265358979323846264f
#define LM_INVPI   0.31830988618379067153777f
#define LM_TWOPI   6.28318530717958647692528f
#define LM_EPSILON 0.0001f      // the EPSILON macro seen by the reference's kernel bodies (bsdf_math.cuh:39-41)

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

// ---- microfacet distributions -----------------------------------------------------------------------------
LM_HD void lm_alpha_from_roughness(float roughness, float anisotropy, float& ax, float& ay)
{
    const float sq = roughness * roughness;
    const float aspect = sqrtf(1.0f + anisotropy * (anisotropy < 0 ? 0.9f : -0.9f));
    ax = fmaxf(0.001f, sq / aspect);
    ay = fmaxf(0.001f, sq * aspect);
}
LM_HD float lm_ggx_D(const lf3& m, float ax, float ay)
{
    if (m.z == 0) return sqrf(ax) * LM_INVPI;
    const float c2 = sqrf(m.z);
    const float st = sqrtf(fmaxf(0.0f, 1 - c2));
    const float tan2 = (1.0f - c2) / c2;
    float stretched;
    if (ax == ay || st == 0.0f) stretched = 1.0f / sqrf(ax);
    else stretched = sqrf(m.x / (st * ax)) + sqrf(m.y / (st * ay));
    return 1.0f / (LM_PI * ax * ay * sqrf(c2) * sqrf(1.0f + tan2 * stretched));
}
LM_HD float lm_ggx_lambda(const lf3& v, float ax, float ay)
{
    if (v.z == 0) return 0;
    const float c2 = v.z * v.z;
    const float st = sqrtf(fmaxf(0.0f, 1 - c2));
    float projected;
    if (ax == ay || st == 0.0f) projected = ax;
    else projected = sqrtf(sqrf((v.x * ax) / st) + sqrf((v.y * ay) / st));
    const float tan2 = sqrf(st) / c2;
    const float a2rcp = sqrf(projected) * tan2;
    return (-1.0f + sqrtf(1.0f + a2rcp)) * 0.5f;
}
LM_HD float lm_ggx_G(const lf3& wi, const lf3& wo, float ax, float ay) { return 1.0f / (1.0f + lm_ggx_lambda(wo, ax, ay) + lm_ggx_lambda(wi, ax, ay)); }
LM_HD float lm_ggx_G1(const lf3& v, float ax, float ay) { return 1.0f / (1.0f + lm_ggx_lambda(v, ax, ay)); }
LM_HD float lm_ggx_pdf(const lf3& v, const lf3& m, float ax, float ay)
{
    if (v.z == 0.0f) return 0;
    return lm_ggx_G1(v, ax, ay) * fabsf(dot3(v, m)) * lm_ggx_D(m, ax, ay) / fabsf(v.z);
}
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
LM_HD float lm_gtr1_D(const lf3& m, float ax)
{
    const float alpha = clampf(ax, 0.001f, 0.999f);
    const float a2 = sqrf(alpha);
    const float a = (a2 - 1.0f) / (LM_PI * lm_logf(a2));
    const float b = (1 / (1 + (a2 - 1) * sqrf(m.z)));
    return a * b;
}
LM_HD float lm_gtr1_lambda(const lf3& v, float ax)
{
    if (v.z == 0) return 0;
    const float c2 = sqrf(v.z);
    const float st = sqrtf(fmaxf(0.0f, 1.0f - c2));
    if (st == 0) return 0;
    const float cot2 = c2 / sqrf(st);
    const float cot = sqrtf(cot2);
    const float a2 = sqrf(clampf(ax, 0.001f, 0.999f));
    const float a = sqrtf(cot2 + a2);
    const float b = sqrtf(cot2 + 1.0f);
    const float c = lm_logf(cot + b);
    const float d = lm_logf(cot + a);
    return (a - b + cot * (c - d)) / (cot * lm_logf(a2));
}
LM_HD float lm_gtr1_G(const lf3& wi, const lf3& wo, float ax) { return 1.0f / (1.0f + lm_gtr1_lambda(wo, ax) + lm_gtr1_lambda(wi, ax)); }
LM_HD lf3 lm_gtr1_sample(float r0, float r1, float ax)
{
    const float alpha = clampf(ax, 0.001f, 0.999f);
    const float a2 = sqrf(alpha);
    const float c2 = (1.0f - lm_powf(a2, 1.0f - r0)) / (1.0f - a2);
    const float st = sqrtf(fmaxf(0.0f, 1.0f - c2));
    float cphi, sphi;
    const float phi = LM_TWOPI * r1;
    lm_sincosf(phi, &sphi, &cphi);
    return v3(cphi * st, sphi * st, sqrtf(c2));
}
LM_HD float lm_gtr1_pdf(const lf3& m, float ax) { return lm_gtr1_D(m, ax) * fabsf(m.z); }

// ---- rough dielectric helpers -------------------------------------------------------------------------------
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
LM_HD lf3 lm_refracted_direction(const lf3& wo, const lf3& m, float cos_wom, float ct, float rcp_eta)
{
    const lf3 wi = cos_wom > 0 ? (rcp_eta * cos_wom - ct) * m - rcp_eta * wo
                               : (rcp_eta * cos_wom + ct) * m - rcp_eta * wo;
    return wi * ((3 - dot3(wi, wi)) * 0.5f);
}
LM_HD float lm_choose_reflection_probability(float F)
{
    const float r = F * 1.f, t = (1 - F) * 1.f, sum = r + t;
    return sum != 0 ? r / sum : 1;
}
LM_HD lf3 lm_half_reflection(const lf3& wo, const lf3& wi) { const lf3 h = normalize3(wi + wo); return h.z < 0 ? (h * -1.f) : h; }
LM_HD lf3 lm_half_refraction(const lf3& wo, const lf3& wi, float eta) { const lf3 h = normalize3(wo + eta * wi); return h.z < 0 ? (h * -1.f) : h; }
LM_HD lf3 lm_eval_reflection(const lf3& color, const lf3& wo, const lf3& wi, const lf3& m, float ax, float ay, float F)
{
    const float denom = fabsf(4 * wo.z * wi.z);
    if (denom == 0) return v3(0);
    const float D = lm_ggx_D(m, ax, ay), G = lm_ggx_G(wi, wo, ax, ay);
    return color * (F * D * G / denom);
}
LM_HD lf3 lm_eval_refraction(float eta, const lf3& color, bool adjoint, const lf3& wo, const lf3& wi, const lf3& m, float ax, float ay, float T)
{
    if (wo.z == 0 || wi.z == 0) return v3(0);
    const float cih = dot3(m, wi), coh = dot3(m, wo);
    const float dots = (cih * coh) / (wi.z * wo.z);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return v3(0);
    const float D = lm_ggx_D(m, ax, ay), G = lm_ggx_G(wi, wo, ax, ay);
    float mult = fabsf(dots) * T * D * G / sqrf(sd);
    if (!adjoint) mult *= sqrf(eta);
    return color * mult;
}
LM_HD float lm_reflection_jacobian(float coh) { return coh == 0 ? 0 : 1 / (4 * fabsf(coh)); }
LM_HD float lm_refraction_jacobian(const lf3& wo, const lf3& wi, const lf3& m, float eta)
{
    const float cih = dot3(m, wi), coh = dot3(m, wo);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return 0;
    return fabsf(cih) * sqrf(eta / sd);
}

// ---- Disney components ----------------------------------------------------------------------------------------
LM_HD float lm_schlick(float u) { const float m = saturatef(1.0f - u), m2 = sqrf(m), m4 = sqrf(m2); return m4 * m; }
LM_HD lf3 lm_mix_spectra(const lf3& a, const lf3& b, float t) { return (1.0f - t) * a + t * b; }
LM_HD lf3 lm_mix_one_with(const lf3& b, float t) { return (1.0f - t) + t * b; }
LM_HD lf3 lm_mix_with_one(const lf3& a, float t) { return (1.0f - t) * a + t; }
LM_HD float lm_clearcoat_roughness(const LmMaterial& sd) { return lerpf(0.1f, 0.001f, LM_P_CLEARCOATGLOSS(sd)); }
LM_HD lf3 lm_specular_fresnel(const LmMaterial& sd, const lf3& o, const lf3& h)
{
    lf3 v = lm_mix_one_with(v3(sd.tint), LM_P_SPECTINT(sd));
    v = v * (LM_P_SPECULAR(sd) * 0.08f);
    v = lm_mix_spectra(v, v3(sd.color), LM_P_METALLIC(sd));
    const float coh = fabsf(dot3(o, h));
    return lm_mix_with_one(v, lm_schlick(coh));
}
LM_HD lf3 lm_clearcoat_fresnel(const LmMaterial& sd, const lf3& o, const lf3& h)
{
    const float coh = fabsf(dot3(o, h));
    return v3(lerpf(0.04f, 1.0f, lm_schlick(coh)) * 0.25f * LM_P_CLEARCOAT(sd));
}
template <bool GGX> LM_HD float lm_mdf_D(const lf3& m, float ax, float ay) { return GGX ? lm_ggx_D(m, ax, ay) : lm_gtr1_D(m, ax); }
template <bool GGX> LM_HD float lm_mdf_G(const lf3& wi, const lf3& wo, float ax, float ay) { return GGX ? lm_ggx_G(wi, wo, ax, ay) : lm_gtr1_G(wi, wo, ax); }
template <bool GGX> LM_HD float lm_mdf_pdf(const lf3& v, const lf3& m, float ax, float ay) { return GGX ? lm_ggx_pdf(v, m, ax, ay) : lm_gtr1_pdf(m, ax); }

template <bool GGX>
LM_HD void lm_sample_mf(const LmMaterial& sd, float r0, float r1, float ax, float ay, const lf3& wol, lf3& wil, float& pdf, lf3& value)
{
    if (wol.z == 0) { value = v3(0); pdf = 0; return; }
    const lf3 m = GGX ? lm_ggx_sample(wol, r0, r1, ax, ay) : lm_gtr1_sample(r0, r1, ax);
    wil = reflect3(wol * -1.0f, m);
    if (wil.z == 0) return;
    const float coh = dot3(wol, m);
    pdf = lm_mdf_pdf<GGX>(wol, m, ax, ay) / fabsf(4.0f * coh);
    if (pdf < 1.0e-6f) return;
    const float D = lm_mdf_D<GGX>(m, ax, ay);
    const float G = lm_mdf_G<GGX>(wil, wol, ax, ay);
    value = GGX ? lm_specular_fresnel(sd, wol, m) : lm_clearcoat_fresnel(sd, wol, m);
    value = value * (D * G);
}
template <bool GGX>
LM_HD float lm_evaluate_mf(const LmMaterial& sd, float ax, float ay, const lf3& wol, const lf3& wil, const lf3& m, lf3& bsdf)
{
    if (wol.z == 0 || wil.z == 0) return 0;
    const float coh = dot3(wol, m);
    if (coh == 0) return 0;
    const float D = lm_mdf_D<GGX>(m, ax, ay);
    const float G = lm_mdf_G<GGX>(wil, wol, ax, ay);
    bsdf = GGX ? lm_specular_fresnel(sd, wol, m) : lm_clearcoat_fresnel(sd, wol, m);
    bsdf = bsdf * (D * G / fabsf(4.0f * wol.z * wil.z));
    return lm_mdf_pdf<GGX>(wol, m, ax, ay) / fabsf(4.0f * coh);
}
LM_HD float lm_evaluate_diffuse(const LmMaterial& sd, const lf3& iN, const lf3& wow, const lf3& wiw, const lf3& m, lf3& value)
{
    const float con = dot3(iN, wow), cin = dot3(iN, wiw), cih = dot3(wiw, m);
    const float fl = lm_schlick(cin), fv = lm_schlick(con);
    const float subsurface = LM_P_SUBSURFACE(sd), rough = LM_P_ROUGHNESS(sd);
    float fd = 0;
    if (subsurface != 1.0f) {
        const float fd90 = 0.5f + 2.0f * sqrf(cih) * rough;
        fd = lerpf(1.f, fd90, fl) * lerpf(1.f, fd90, fv);
    }
    if (subsurface > 0) {
        const float fss90 = sqrf(cih) * rough;
        const float fss = lerpf(1.0f, fss90, fl) * lerpf(1.0f, fss90, fv);
        const float ss = 1.25f * (fss * (1.0f / (fabsf(con) + fabsf(cin)) - 0.5f) + 0.5f);
        fd = lerpf(fd, ss, subsurface);
    }
    value = v3(sd.color) * fd * LM_INVPI * (1.0f - LM_P_METALLIC(sd));
    return fabsf(cin) * LM_INVPI;
}
LM_HD float lm_evaluate_sheen(const LmMaterial& sd, const lf3& wiw, const lf3& m, lf3& value)
{
    const float cih = dot3(wiw, m);
    const float fh = lm_schlick(cih);
    value = lm_mix_one_with(v3(sd.tint), LM_P_SHEENTINT(sd));
    value = value * (fh * LM_P_SHEEN(sd) * (1.0f - LM_P_METALLIC(sd)));
    return 1.0f / (2 * LM_PI);
}
LM_HD lf3 lm_w2t(const lf3& V, const lf3& N, const lf3& T, const lf3& B) { return v3(dot3(V, T), dot3(V, B), dot3(V, N)); }
LM_HD lf3 lm_t2w(const lf3& V, const lf3& N, const lf3& T, const lf3& B) { return V.x * T + V.y * B + V.z * N; }
LM_HD void lm_component_weights(const LmMaterial& sd, float& w0, float& w1, float& w2, float& w3)
{
    const float metallic = LM_P_METALLIC(sd);
    w0 = lerpf(sd.tint.w, 0.f, metallic);
    w1 = lerpf(LM_P_SHEEN(sd), 0.f, metallic);
    w2 = lerpf(LM_P_SPECULAR(sd), 1.f, metallic);
    w3 = LM_P_CLEARCOAT(sd) * 0.25f;
    const float inv = 1.0f / (w0 + w1 + w2 + w3);
    w0 *= inv; w1 *= inv; w2 *= inv; w3 *= inv;
}

// ---- sampling (reference: disney.cuh:173-304) ---------------------------------------------------------------------
LM_HD lf3 lm_sample_bsdf(const LmMaterial& sd, lf3 iN, const lf3& N, const lf3& iT, const lf3& wow, float distance,
                         float r0, float r1, float r2, lf3& wiw, float& pdf, bool& specular)
{
    const float flip = (dot3(wow, N) < 0) ? -1.f : 1.f;
    iN = iN * flip;
    const lf3 B = normalize3(cross3(iN, iT));
    const lf3 T = normalize3(cross3(iN, B));
    const float transmission = LM_P_TRANSMISSION(sd);
    const float rough = LM_P_ROUGHNESS(sd), aniso = LM_P_ANISOTROPIC(sd);
    if (r0 < transmission) {
        specular = true;
        const float r3 = r0 / transmission;
        const lf3 wol = lm_w2t(wow, iN, T, B);
        const float ior = sd.transmittance.w;
        const float eta = flip < 0 ? (1 / ior) : ior;
        if (eta == 1) return v3(0);
        const lf3 beer = v3(lm_expf(-sd.transmittance.x * distance * 2.0f),
                            lm_expf(-sd.transmittance.y * distance * 2.0f),
                            lm_expf(-sd.transmittance.z * distance * 2.0f));
        float ax, ay;
        lm_alpha_from_roughness(rough, aniso, ax, ay);
        const lf3 m = lm_ggx_sample(wol, r1, r3, ax, ay);
        const float rcp_eta = 1 / eta, cos_wom = clampf(dot3(wol, m), -1.0f, 1.0f);
        float ct, jacobian;
        const float F = lm_fresnel_reflectance(cos_wom, eta, ct);
        lf3 wil, ret;
        if (r2 < F) {
            wil = reflect3(wol * -1.0f, m);
            if (wil.z * wol.z <= 0) return v3(0);
            ret = lm_eval_reflection(v3(sd.color), wol, wil, m, ax, ay, F);
            pdf = F; jacobian = lm_reflection_jacobian(cos_wom);
        } else {
            wil = lm_refracted_direction(wol, m, cos_wom, ct, eta);        // eta where rcp_eta is expected: reference behaviour (disney.cuh:219)
            if (wil.z * wol.z > 0) return v3(0);
            ret = lm_eval_refraction(rcp_eta, v3(sd.color), false, wol, wil, m, ax, ay, 1 - F);
            pdf = 1 - F; jacobian = lm_refraction_jacobian(wol, wil, m, rcp_eta);
        }
        pdf *= jacobian * lm_ggx_pdf(wol, m, ax, ay);
        if (pdf > 1.0e-6f) wiw = lm_t2w(wil, iN, T, B);
        return ret * beer;
    }
    const float r3 = (r0 - transmission) / (1 - transmission);
    float w0, w1, w2, w3;
    lm_component_weights(sd, w0, w1, w2, w3);
    const float cdfx = w0, cdfy = w0 + w1, cdfz = w0 + w1 + w2;
    float probability = 0.f, component_pdf = 0.f;
    lf3 contrib = v3(0), value = v3(0);
    if (r3 < cdfy) {
        const float rr = r3 / cdfy;
        {
            const float term1 = LM_TWOPI * rr, term2 = sqrtf(1 - r1);
            float s, c;
            lm_sincosf(term1, &s, &c);
            wiw = (c * term2 * T) + (s * term2) * B + sqrtf(r1) * iN;
        }
        const lf3 m = normalize3(wiw + wow);
        if (r3 < cdfx) { component_pdf = lm_evaluate_diffuse(sd, iN, wow, wiw, m, value); probability = w0 * component_pdf; w0 = 0; }
        else { component_pdf = lm_evaluate_sheen(sd, wiw, m, value); probability = w1 * component_pdf; w1 = 0; }
    } else {
        const lf3 wol = lm_w2t(wow, iN, T, B);
        lf3 wil = v3(0);
        if (r3 < cdfz) {
            const float rr = (r3 - cdfy) / (cdfz - cdfy);
            float ax, ay;
            lm_alpha_from_roughness(rough, aniso, ax, ay);
            lm_sample_mf<true>(sd, rr, r1, ax, ay, wol, wil, component_pdf, value);
            probability = w2 * component_pdf; w2 = 0;
        } else {
            const float rr = (r3 - cdfz) / (1 - cdfz);
            const float alpha = lm_clearcoat_roughness(sd);
            lm_sample_mf<false>(sd, rr, r1, alpha, alpha, wol, wil, component_pdf, value);
            probability = w3 * component_pdf; w3 = 0;
        }
        value = value * (1.0f / fabsf(4.0f * wol.z * wil.z));
        wiw = lm_t2w(wil, iN, T, B);
    }
    if (w0 + w1 > 0) {
        const lf3 m = normalize3(wiw + wow);
        if (w0 > 0) { contrib = v3(0); probability += w0 * lm_evaluate_diffuse(sd, iN, wow, wiw, m, contrib); value = value + contrib; }
        if (w1 > 0) { contrib = v3(0); probability += w1 * lm_evaluate_sheen(sd, wiw, m, contrib); value = value + contrib; }
    }
    if (w2 + w3 > 0) {
        const lf3 wol = lm_w2t(wow, iN, T, B);
        const lf3 wil = lm_w2t(wiw, iN, T, B);
        const lf3 m = normalize3(wol + wil);
        if (w2 > 0) {
            float ax, ay;
            lm_alpha_from_roughness(rough, aniso, ax, ay);
            contrib = v3(0);
            probability += w2 * lm_evaluate_mf<true>(sd, ax, ay, wol, wil, m, contrib);
            value = value + contrib;
        }
        if (w3 > 0) {
            const float alpha = lm_clearcoat_roughness(sd);
            contrib = v3(0);
            probability += w3 * lm_evaluate_mf<false>(sd, alpha, alpha, wol, wil, m, contrib);
            value = value + contrib;
        }
    }
    if (probability > 1.0e-6f) pdf = probability; else pdf = 0;
    return value;
}

// ---- evaluation (reference: disney.cuh:320-405) --------------------------------------------------------------------
LM_HD lf3 lm_evaluate_bsdf(const LmMaterial& sd, const lf3& iN, const lf3& iT, const lf3& wow, const lf3& wiw, float& pdf)
{
    lf3 stBSDF = v3(0);
    float stPDF = 0.f;
    const float transmission = LM_P_TRANSMISSION(sd);
    const float rough = LM_P_ROUGHNESS(sd), aniso = LM_P_ANISOTROPIC(sd);
    if (transmission > 0.f) {
        const lf3 B = normalize3(cross3(iN, iT));
        const lf3 T = normalize3(cross3(iN, B));
        const lf3 wol = lm_w2t(wow, iN, T, B);
        const lf3 wil = lm_w2t(wiw, iN, T, B);
        const float ior = sd.transmittance.w;
        const float eta = wol.z > 0 ? ior : (1.0f / ior);
        if (eta == 1) { pdf = 0; return v3(0); }
        float ax, ay, jacobian;
        lm_alpha_from_roughness(rough, aniso, ax, ay);
        lf3 m;
        if (wil.z * wol.z >= 0) {
            m = lm_half_reflection(wol, wil);
            const float cos_wom = dot3(wol, m);
            float ct;
            const float F = lm_fresnel_reflectance(cos_wom, 1 / eta, ct);
            stBSDF = lm_eval_reflection(v3(sd.color), wol, wil, m, ax, ay, F);
            stPDF = lm_choose_reflection_probability(F);
            jacobian = lm_reflection_jacobian(cos_wom);
        } else {
            m = lm_half_refraction(wol, wil, eta);
            const float cos_wom = dot3(wol, m);
            float ct;
            const float F = lm_fresnel_reflectance(cos_wom, 1 / eta, ct);
            stBSDF = lm_eval_refraction(eta, v3(sd.color), false, wol, wil, m, ax, ay, 1 - F);
            stPDF = 1 - lm_choose_reflection_probability(F);
            jacobian = lm_refraction_jacobian(wol, wil, m, eta);
        }
        stPDF *= jacobian * lm_ggx_pdf(wol, m, ax, ay);
    }
    if (rough <= 0.001f) { pdf = stPDF; return stBSDF; }
    const lf3 B = normalize3(cross3(iN, iT));
    const lf3 T = normalize3(cross3(iN, B));
    float w0, w1, w2, w3;
    lm_component_weights(sd, w0, w1, w2, w3);
    pdf = 0;
    lf3 value = v3(0);
    if (w0 + w1 > 0) {
        const lf3 m = normalize3(wiw + wow);
        if (w0 > 0) pdf += w0 * lm_evaluate_diffuse(sd, iN, wow, wiw, m, value);
        if (w1 > 0) pdf += w1 * lm_evaluate_sheen(sd, wiw, m, value);     // overwrites the diffuse value: reference behaviour (disney.cuh:373)
    }
    if (w2 + w3 > 0) {
        const lf3 wol = lm_w2t(wow, iN, T, B);
        const lf3 wil = lm_w2t(wiw, iN, T, B);
        const lf3 m = normalize3(wol + wil);
        if (w2 > 0) {
            float ax, ay;
            lm_alpha_from_roughness(rough, aniso, ax, ay);
            lf3 contrib = v3(0);
            const float p = lm_evaluate_mf<true>(sd, ax, ay, wol, wil, m, contrib);
            if (p > 0) { pdf += w2 * p; value = value + contrib; }
        }
        if (w3 > 0) {
            const float alpha = lm_clearcoat_roughness(sd);
            lf3 contrib = v3(0);
            const float p = lm_evaluate_mf<false>(sd, alpha, alpha, wol, wil, m, contrib);
            if (p > 0) { pdf += w3 * p; value = value + contrib; }
        }
    }
    pdf = (pdf * (1.f - transmission));
    pdf += (stPDF * transmission);
    return (stBSDF * transmission) + (value * (1.f - transmission));
}
