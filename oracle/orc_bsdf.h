// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h).
//
// orc_bsdf.h: packed material record and the Disney principled BSDF, restated from
//   LumenPT/src/Shaders/CppCommon/MaterialStructs.h:13-261   (8-bit parameter packing)
//   LumenPT/src/CUDAKernels/ggxmdf.cuh:43-228                 (GGX / GTR1 microfacet distributions)
//   LumenPT/src/CUDAKernels/frosted.cuh:28-120                (rough dielectric helpers)
//   LumenPT/src/CUDAKernels/disney.cuh:33-150,173-304,320-405 (components, SampleBSDF, EvaluateBSDF)
//   LumenPT/src/CUDAKernels/bsdf_math.cuh:57-149              (tangent frames, cosine sampling)
// The reference compiles the __CUDACC__ branches (sincosf forms); those are the ones followed here.
// Where the reference reads an uninitialised local (component_pdf / contrib / wil after an early return in
// sample_mf / evaluate_mf, disney.cuh:97,104,117,119) the oracle defines the value as zero.
#pragma once
#include "orc_math.h"

namespace orc {

static constexpr float kPI = 3.14159265358979323846264f;
static constexpr float kINVPI = 0.31830988618379067153777f;
static constexpr float kTWOPI = 6.28318530717958647692528f;
static constexpr float kBsdfEpsilon = 0.0001f;   // the EPSILON *macro* of bsdf_math.cuh:39-41

// ---- MaterialStructs.h:13-29: four float4 + uint4 of byte-packed parameters (80 bytes) -------------------
struct Material {
    f4 color;          // albedo, w = alpha
    f4 emissive;
    f4 transmittance;  // w = refractive index slot (holds eta = 1/ior after extraction)
    f4 tint;           // w = luminance
    uint32_t params[4];
};
enum ParamSlot {       // (word, shift) pairs of MaterialStructs.h:84-217
    P_METALLIC = 0x00, P_SUBSURFACE = 0x08, P_SPECULAR = 0x10, P_ROUGHNESS = 0x18,
    P_SPECTINT = 0x20, P_ANISOTROPIC = 0x28, P_SHEEN = 0x30, P_SHEENTINT = 0x38,
    P_CLEARCOAT = 0x40, P_CLEARCOATGLOSS = 0x48, P_TRANSMISSION = 0x50
};
static inline void mat_set(Material& m, ParamSlot slot, float v)
{
    const unsigned word = (unsigned)slot >> 5, shift = (unsigned)slot & 31u;
    const uint32_t q = (uint32_t)(v * 255.f);                 // truncation, MaterialStructs.h:86
    m.params[word] &= ~(255u << shift);
    m.params[word] |= q << shift;
}
static inline float mat_get(const Material& m, ParamSlot slot)
{
    const unsigned word = (unsigned)slot >> 5, shift = (unsigned)slot & 31u;
    return (float)((m.params[word] >> shift) & 255u) * (1.0f / 255.0f);   // CHAR_TO_FLOAT, MaterialStructs.h:11
}
static inline Material mat_zero()
{
    Material m; memset(&m, 0, sizeof m); return m;
}

// ---- ggxmdf.cuh ------------------------------------------------------------------------------------------
static inline void alpha_from_roughness(float roughness, float anisotropy, float& ax, float& ay)   // :221-227
{
    const float sq = roughness * roughness;
    const float aspect = sqrtf(1.0f + anisotropy * (anisotropy < 0 ? 0.9f : -0.9f));
    ax = fmaxf(0.001f, sq / aspect);
    ay = fmaxf(0.001f, sq * aspect);
}
static inline float ggx_D(const f3& m, float ax, float ay)                                           // :43-53
{
    if (m.z == 0) return sqr(ax) * kINVPI;
    const float c2 = sqr(m.z);
    const float st = sqrtf(fmaxf(0.0f, 1 - c2));
    const float tan2 = (1.0f - c2) / c2;
    float stretched;
    if (ax == ay || st == 0.0f) stretched = 1.0f / sqr(ax);
    else stretched = sqr(m.x / (st * ax)) + sqr(m.y / (st * ay));
    return 1.0f / (kPI * ax * ay * sqr(c2) * sqr(1.0f + tan2 * stretched));
}
static inline float ggx_lambda(const f3& v, float ax, float ay)                                      // :55-66
{
    if (v.z == 0) return 0;
    const float c2 = v.z * v.z;
    const float st = sqrtf(fmaxf(0.0f, 1 - c2));
    float projected;
    if (ax == ay || st == 0.0f) projected = ax;
    else projected = sqrtf(sqr((v.x * ax) / st) + sqr((v.y * ay) / st));
    const float tan2 = sqr(st) / c2;
    const float a2rcp = sqr(projected) * tan2;
    return (-1.0f + sqrtf(1.0f + a2rcp)) * 0.5f;
}
static inline float ggx_G(const f3& wi, const f3& wo, float ax, float ay)                           // :68-71
{
    return 1.0f / (1.0f + ggx_lambda(wo, ax, ay) + ggx_lambda(wi, ax, ay));
}
static inline float ggx_G1(const f3& v, float ax, float ay) { return 1.0f / (1.0f + ggx_lambda(v, ax, ay)); }
static inline float ggx_pdf(const f3& v, const f3& m, float ax, float ay)                           // :155-165
{
    if (v.z == 0.0f) return 0;
    return ggx_G1(v, ax, ay) * fabsf(dot(v, m)) * ggx_D(m, ax, ay) / fabsf(v.z);
}
static inline f3 ggx_sample(const f3& v, float r0, float r1, float ax, float ay)                    // :78-106
{
    const float sgn = v.z < 0.0f ? -1.0f : 1.0f;
    const f3 stretched = normalize(mk3(sgn * v.x * ax, sgn * v.y * ay, sgn * v.z));
    const f3 t1 = v.z < 0.9999f ? normalize(cross(stretched, mk3(0, 0, 1))) : mk3(1, 0, 0);
    const f3 t2 = cross(t1, stretched);
    const float a = 1.0f / (1.0f + stretched.z);
    const float r = sqrtf(r0);
    const float phi = r1 < a ? (r1 / a * kPI) : (kPI + (r1 - a) / (1.0f - a) * kPI);
    float p1, p2;
    det_sincosf(phi, &p2, &p1);
    p1 *= r;
    p2 *= r * (r1 < a ? 1.0f : stretched.z);
    const f3 h = p1 * t1 + p2 * t2 + sqrtf(fmaxf(0.0f, 1.0f - p1 * p1 - p2 * p2)) * stretched;
    return normalize(mk3(h.x * ax, h.y * ay, fmaxf(0.0f, h.z)));
}
static inline float gtr1_D(const f3& m, float ax)                                                    // :174-181
{
    const float alpha = clampf(ax, 0.001f, 0.999f);
    const float a2 = sqr(alpha);
    const float a = (a2 - 1.0f) / (kPI * det_logf(a2));
    const float b = (1 / (1 + (a2 - 1) * sqr(m.z)));
    return a * b;
}
static inline float gtr1_lambda(const f3& v, float ax)                                               // :183-199
{
    if (v.z == 0) return 0;
    const float c2 = sqr(v.z);
    const float st = sqrtf(fmaxf(0.0f, 1.0f - c2));
    if (st == 0) return 0;
    const float cot2 = c2 / sqr(st);
    const float cot = sqrtf(cot2);
    const float a2 = sqr(clampf(ax, 0.001f, 0.999f));
    const float a = sqrtf(cot2 + a2);
    const float b = sqrtf(cot2 + 1.0f);
    const float c = det_logf(cot + b);
    const float d = det_logf(cot + a);
    return (a - b + cot * (c - d)) / (cot * det_logf(a2));
}
static inline float gtr1_G(const f3& wi, const f3& wo, float ax)                                     // :201-204
{
    return 1.0f / (1.0f + gtr1_lambda(wo, ax) + gtr1_lambda(wi, ax));
}
static inline f3 gtr1_sample(float r0, float r1, float ax)                                           // :211-224
{
    const float alpha = clampf(ax, 0.001f, 0.999f);
    const float a2 = sqr(alpha);
    const float c2 = (1.0f - det_powf(a2, 1.0f - r0)) / (1.0f - a2);
    const float st = sqrtf(fmaxf(0.0f, 1.0f - c2));
    float cphi, sphi;
    const float phi = kTWOPI * r1;
    det_sincosf(phi, &sphi, &cphi);
    return mk3(cphi * st, sphi * st, sqrtf(c2));                                                     // make_unit_vector :30-33
}
static inline float gtr1_pdf(const f3& m, float ax) { return gtr1_D(m, ax) * fabsf(m.z); }          // :226-229

// ---- frosted.cuh -----------------------------------------------------------------------------------------
static inline float fresnel_dielectric(float eta, float ci, float ct)                                // :28-33
{
    if (ci == 0 && ct == 0) return 1;
    const float k0 = eta * ct, k1 = eta * ci;
    return 0.5f * (sqr((ci - k0) / (ci + k0)) + sqr((ct - k1) / (ct + k1)));
}
static inline float fresnel_reflectance(float ci, float eta, float& ct)                              // :35-43
{
    const float st2 = (1 - sqr(ci)) * sqr(eta);
    if (st2 > 1) { ct = 0; return 1; }
    ct = fminf(sqrtf(fmaxf(1 - st2, 0.0f)), 1.0f);
    return fresnel_dielectric(eta, fabsf(ci), ct);
}
static inline f3 refracted_direction(const f3& wo, const f3& m, float cos_wom, float ct, float rcp_eta)   // :56-62
{
    const f3 wi = cos_wom > 0 ? (rcp_eta * cos_wom - ct) * m - rcp_eta * wo
                              : (rcp_eta * cos_wom + ct) * m - rcp_eta * wo;
    return wi * ((3 - dot(wi, wi)) * 0.5f);                                                          // improve_normalization :51-54
}
static inline float choose_reflection_probability(float F)                                           // :64-70 with weights 1,1
{
    const float r = F * 1.f, t = (1 - F) * 1.f, sum = r + t;
    return sum != 0 ? r / sum : 1;
}
static inline f3 half_reflection(const f3& wo, const f3& wi) { const f3 h = normalize(wi + wo); return h.z < 0 ? (h * -1.f) : h; }          // :72-76
static inline f3 half_refraction(const f3& wo, const f3& wi, float eta) { const f3 h = normalize(wo + eta * wi); return h.z < 0 ? (h * -1.f) : h; }  // :87-91
static inline f3 eval_reflection(const f3& color, const f3& wo, const f3& wi, const f3& m, float ax, float ay, float F)   // :78-85
{
    const float denom = fabsf(4 * wo.z * wi.z);
    if (denom == 0) return mk3(0);
    const float D = ggx_D(m, ax, ay), G = ggx_G(wi, wo, ax, ay);
    return color * (F * D * G / denom);
}
static inline f3 eval_refraction(float eta, const f3& color, bool adjoint, const f3& wo, const f3& wi, const f3& m, float ax, float ay, float T)  // :93-107
{
    if (wo.z == 0 || wi.z == 0) return mk3(0);
    const float cih = dot(m, wi), coh = dot(m, wo);
    const float dots = (cih * coh) / (wi.z * wo.z);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return mk3(0);
    const float D = ggx_D(m, ax, ay), G = ggx_G(wi, wo, ax, ay);
    float mult = fabsf(dots) * T * D * G / sqr(sd);
    if (!adjoint) mult *= sqr(eta);
    return color * mult;
}
static inline float reflection_jacobian(float coh) { return coh == 0 ? 0 : 1 / (4 * fabsf(coh)); }   // :109-113
static inline float refraction_jacobian(const f3& wo, const f3& wi, const f3& m, float eta)          // :115-121
{
    const float cih = dot(m, wi), coh = dot(m, wo);
    const float sd = coh + eta * cih;
    if (fabsf(sd) < 1.0e-6f) return 0;
    return fabsf(cih) * sqr(eta / sd);
}

// ---- disney.cuh:33-150 -----------------------------------------------------------------------------------
static inline float schlick(float u) { const float m = saturatef(1.0f - u), m2 = sqr(m), m4 = sqr(m2); return m4 * m; }
static inline f3 mix_spectra(const f3& a, const f3& b, float t) { return (1.0f - t) * a + t * b; }
static inline f3 mix_one_with(const f3& b, float t) { return (1.0f - t) + t * b; }
static inline f3 mix_with_one(const f3& a, float t) { return (1.0f - t) * a + t; }
static inline float clearcoat_roughness(const Material& sd) { return lerpf(0.1f, 0.001f, mat_get(sd, P_CLEARCOATGLOSS)); }
static inline f3 specular_fresnel(const Material& sd, const f3& o, const f3& h)
{
    f3 v = mix_one_with(mk3(sd.tint), mat_get(sd, P_SPECTINT));
    v *= mat_get(sd, P_SPECULAR) * 0.08f;
    v = mix_spectra(v, mk3(sd.color), mat_get(sd, P_METALLIC));
    const float coh = fabsf(dot(o, h));
    return mix_with_one(v, schlick(coh));
}
static inline f3 clearcoat_fresnel(const Material& sd, const f3& o, const f3& h)
{
    const float coh = fabsf(dot(o, h));
    return mk3(lerpf(0.04f, 1.0f, schlick(coh)) * 0.25f * mat_get(sd, P_CLEARCOAT));
}
enum Mdf { MDF_GGX, MDF_GTR1 };
static inline float mdf_D(Mdf k, const f3& m, float ax, float ay) { return k == MDF_GGX ? ggx_D(m, ax, ay) : gtr1_D(m, ax); }
static inline float mdf_G(Mdf k, const f3& wi, const f3& wo, float ax, float ay) { return k == MDF_GGX ? ggx_G(wi, wo, ax, ay) : gtr1_G(wi, wo, ax); }
static inline float mdf_pdf(Mdf k, const f3& v, const f3& m, float ax, float ay) { return k == MDF_GGX ? ggx_pdf(v, m, ax, ay) : gtr1_pdf(m, ax); }

static inline void sample_mf(Mdf k, const Material& sd, float r0, float r1, float ax, float ay, const f3& wol,
                             f3& wil, float& pdf, f3& value)                                         // :79-99
{
    if (wol.z == 0) { value = mk3(0); pdf = 0; return; }
    const f3 m = k == MDF_GGX ? ggx_sample(wol, r0, r1, ax, ay) : gtr1_sample(r0, r1, ax);
    wil = reflect(wol * -1.0f, m);
    if (wil.z == 0) return;
    const float coh = dot(wol, m);
    pdf = mdf_pdf(k, wol, m, ax, ay) / fabsf(4.0f * coh);
    if (pdf < 1.0e-6f) return;
    const float D = mdf_D(k, m, ax, ay);
    const float G = mdf_G(k, wil, wol, ax, ay);
    value = k == MDF_GGX ? specular_fresnel(sd, wol, m) : clearcoat_fresnel(sd, wol, m);
    value *= D * G;
}
static inline float evaluate_mf(Mdf k, const Material& sd, float ax, float ay, const f3& wol, const f3& wil, const f3& m, f3& bsdf)   // :101-114
{
    if (wol.z == 0 || wil.z == 0) return 0;
    const float coh = dot(wol, m);
    if (coh == 0) return 0;
    const float D = mdf_D(k, m, ax, ay);
    const float G = mdf_G(k, wil, wol, ax, ay);
    bsdf = k == MDF_GGX ? specular_fresnel(sd, wol, m) : clearcoat_fresnel(sd, wol, m);
    bsdf *= D * G / fabsf(4.0f * wol.z * wil.z);
    return mdf_pdf(k, wol, m, ax, ay) / fabsf(4.0f * coh);
}
static inline float evaluate_diffuse(const Material& sd, const f3& iN, const f3& wow, const f3& wiw, const f3& m, f3& value)   // :116-139
{
    const float con = dot(iN, wow), cin = dot(iN, wiw), cih = dot(wiw, m);
    const float fl = schlick(cin), fv = schlick(con);
    const float subsurface = mat_get(sd, P_SUBSURFACE), rough = mat_get(sd, P_ROUGHNESS);
    float fd = 0;
    if (subsurface != 1.0f) {
        const float fd90 = 0.5f + 2.0f * sqr(cih) * rough;
        fd = lerpf(1.f, fd90, fl) * lerpf(1.f, fd90, fv);
    }
    if (subsurface > 0) {
        const float fss90 = sqr(cih) * rough;
        const float fss = lerpf(1.0f, fss90, fl) * lerpf(1.0f, fss90, fv);
        const float ss = 1.25f * (fss * (1.0f / (fabsf(con) + fabsf(cin)) - 0.5f) + 0.5f);
        fd = lerpf(fd, ss, subsurface);
    }
    value = mk3(sd.color) * fd * kINVPI * (1.0f - mat_get(sd, P_METALLIC));
    return fabsf(cin) * kINVPI;
}
static inline float evaluate_sheen(const Material& sd, const f3& wiw, const f3& m, f3& value)       // :141-150
{
    const float cih = dot(wiw, m);
    const float fh = schlick(cih);
    value = mix_one_with(mk3(sd.tint), mat_get(sd, P_SHEENTINT));
    value *= fh * mat_get(sd, P_SHEEN) * (1.0f - mat_get(sd, P_METALLIC));
    return 1.0f / (2 * kPI);
}

static inline f3 w2t(const f3& V, const f3& N, const f3& T, const f3& B) { return mk3(dot(V, T), dot(V, B), dot(V, N)); }   // bsdf_math.cuh:77-80
static inline f3 t2w(const f3& V, const f3& N, const f3& T, const f3& B) { return V.x * T + V.y * B + V.z * N; }             // bsdf_math.cuh:89-92

static inline void component_weights(const Material& sd, float w[4])
{
    const float metallic = mat_get(sd, P_METALLIC);
    w[0] = lerpf(sd.tint.w, 0.f, metallic);                      // luminance lives in tint.w
    w[1] = lerpf(mat_get(sd, P_SHEEN), 0.f, metallic);
    w[2] = lerpf(mat_get(sd, P_SPECULAR), 1.f, metallic);
    w[3] = mat_get(sd, P_CLEARCOAT) * 0.25f;
    const float inv = 1.0f / (w[0] + w[1] + w[2] + w[3]);
    w[0] *= inv; w[1] *= inv; w[2] *= inv; w[3] *= inv;
}

// ---- disney.cuh:173-304 ----------------------------------------------------------------------------------
static inline f3 sample_bsdf(const Material& sd, f3 iN, const f3& N, const f3& iT, const f3& wow, float distance,
                             float r0, float r1, float r2, f3& wiw, float& pdf, bool& specular)
{
    const float flip = (dot(wow, N) < 0) ? -1.f : 1.f;
    iN *= flip;
    const f3 B = normalize(cross(iN, iT));
    const f3 T = normalize(cross(iN, B));
    const float transmission = mat_get(sd, P_TRANSMISSION);
    const float rough = mat_get(sd, P_ROUGHNESS), aniso = mat_get(sd, P_ANISOTROPIC);
    if (r0 < transmission) {
        specular = true;
        const float r3 = r0 / transmission;
        const f3 wol = w2t(wow, iN, T, B);
        const float ior = sd.transmittance.w;
        const float eta = flip < 0 ? (1 / ior) : ior;
        if (eta == 1) return mk3(0);
        const f3 beer = mk3(det_expf(-sd.transmittance.x * distance * 2.0f),
                            det_expf(-sd.transmittance.y * distance * 2.0f),
                            det_expf(-sd.transmittance.z * distance * 2.0f));
        float ax, ay;
        alpha_from_roughness(rough, aniso, ax, ay);
        const f3 m = ggx_sample(wol, r1, r3, ax, ay);
        const float rcp_eta = 1 / eta, cos_wom = clampf(dot(wol, m), -1.0f, 1.0f);
        float ct, jacobian;
        const float F = fresnel_reflectance(cos_wom, eta, ct);
        f3 wil, ret;
        if (r2 < F) {
            wil = reflect(wol * -1.0f, m);
            if (wil.z * wol.z <= 0) return mk3(0);
            ret = eval_reflection(mk3(sd.color), wol, wil, m, ax, ay, F);
            pdf = F; jacobian = reflection_jacobian(cos_wom);
        } else {
            wil = refracted_direction(wol, m, cos_wom, ct, eta);      // sic: eta passed where rcp_eta is expected (disney.cuh:219)
            if (wil.z * wol.z > 0) return mk3(0);
            ret = eval_refraction(rcp_eta, mk3(sd.color), false, wol, wil, m, ax, ay, 1 - F);
            pdf = 1 - F; jacobian = refraction_jacobian(wol, wil, m, rcp_eta);
        }
        pdf *= jacobian * ggx_pdf(wol, m, ax, ay);
        if (pdf > 1.0e-6f) wiw = t2w(wil, iN, T, B);
        return ret * beer;
    }
    const float r3 = (r0 - transmission) / (1 - transmission);
    float w[4];
    component_weights(sd, w);
    const float cdfx = w[0], cdfy = w[0] + w[1], cdfz = w[0] + w[1] + w[2];
    float probability = 0.f, component_pdf = 0.f;
    f3 contrib = mk3(0), value = mk3(0);
    if (r3 < cdfy) {
        const float rr = r3 / cdfy;
        {   // DiffuseReflectionCosWeighted, bsdf_math.cuh:140-146
            const float term1 = kTWOPI * rr, term2 = sqrtf(1 - r1);
            float s, c;
            det_sincosf(term1, &s, &c);
            wiw = (c * term2 * T) + (s * term2) * B + sqrtf(r1) * iN;
        }
        const f3 m = normalize(wiw + wow);
        if (r3 < cdfx) { component_pdf = evaluate_diffuse(sd, iN, wow, wiw, m, value); probability = w[0] * component_pdf; w[0] = 0; }
        else { component_pdf = evaluate_sheen(sd, wiw, m, value); probability = w[1] * component_pdf; w[1] = 0; }
    } else {
        const f3 wol = w2t(wow, iN, T, B);
        f3 wil = mk3(0);
        if (r3 < cdfz) {
            const float rr = (r3 - cdfy) / (cdfz - cdfy);
            float ax, ay;
            alpha_from_roughness(rough, aniso, ax, ay);
            sample_mf(MDF_GGX, sd, rr, r1, ax, ay, wol, wil, component_pdf, value);
            probability = w[2] * component_pdf; w[2] = 0;
        } else {
            const float rr = (r3 - cdfz) / (1 - cdfz);
            const float alpha = clearcoat_roughness(sd);
            sample_mf(MDF_GTR1, sd, rr, r1, alpha, alpha, wol, wil, component_pdf, value);
            probability = w[3] * component_pdf; w[3] = 0;
        }
        value *= 1.0f / fabsf(4.0f * wol.z * wil.z);
        wiw = t2w(wil, iN, T, B);
    }
    if (w[0] + w[1] > 0) {
        const f3 m = normalize(wiw + wow);
        if (w[0] > 0) { contrib = mk3(0); probability += w[0] * evaluate_diffuse(sd, iN, wow, wiw, m, contrib); value += contrib; }
        if (w[1] > 0) { contrib = mk3(0); probability += w[1] * evaluate_sheen(sd, wiw, m, contrib); value += contrib; }
    }
    if (w[2] + w[3] > 0) {
        const f3 wol = w2t(wow, iN, T, B);
        const f3 wil = w2t(wiw, iN, T, B);
        const f3 m = normalize(wol + wil);
        if (w[2] > 0) {
            float ax, ay;
            alpha_from_roughness(rough, aniso, ax, ay);
            contrib = mk3(0);
            probability += w[2] * evaluate_mf(MDF_GGX, sd, ax, ay, wol, wil, m, contrib);
            value += contrib;
        }
        if (w[3] > 0) {
            const float alpha = clearcoat_roughness(sd);
            contrib = mk3(0);
            probability += w[3] * evaluate_mf(MDF_GTR1, sd, alpha, alpha, wol, wil, m, contrib);
            value += contrib;
        }
    }
    if (probability > 1.0e-6f) pdf = probability; else pdf = 0;
    return value;
}

// ---- disney.cuh:320-405 ----------------------------------------------------------------------------------
static inline f3 evaluate_bsdf(const Material& sd, const f3& iN, const f3& iT, const f3& wow, const f3& wiw, float& pdf)
{
    f3 stBSDF = mk3(0);
    float stPDF = 0.f;
    const float transmission = mat_get(sd, P_TRANSMISSION);
    const float rough = mat_get(sd, P_ROUGHNESS), aniso = mat_get(sd, P_ANISOTROPIC);
    if (transmission > 0.f) {
        const f3 B = normalize(cross(iN, iT));
        const f3 T = normalize(cross(iN, B));
        const f3 wol = w2t(wow, iN, T, B);
        const f3 wil = w2t(wiw, iN, T, B);
        const float ior = sd.transmittance.w;
        const float eta = wol.z > 0 ? ior : (1.0f / ior);
        if (eta == 1) { pdf = 0; return mk3(0); }
        float ax, ay, jacobian;
        alpha_from_roughness(rough, aniso, ax, ay);
        f3 m;
        if (wil.z * wol.z >= 0) {
            m = half_reflection(wol, wil);
            const float cos_wom = dot(wol, m);
            float ct;
            const float F = fresnel_reflectance(cos_wom, 1 / eta, ct);
            stBSDF = eval_reflection(mk3(sd.color), wol, wil, m, ax, ay, F);
            stPDF = choose_reflection_probability(F);
            jacobian = reflection_jacobian(cos_wom);
        } else {
            m = half_refraction(wol, wil, eta);
            const float cos_wom = dot(wol, m);
            float ct;
            const float F = fresnel_reflectance(cos_wom, 1 / eta, ct);
            stBSDF = eval_refraction(eta, mk3(sd.color), false, wol, wil, m, ax, ay, 1 - F);
            stPDF = 1 - choose_reflection_probability(F);
            jacobian = refraction_jacobian(wol, wil, m, eta);
        }
        stPDF *= jacobian * ggx_pdf(wol, m, ax, ay);
    }
    if (rough <= 0.001f) { pdf = stPDF; return stBSDF; }
    const f3 B = normalize(cross(iN, iT));
    const f3 T = normalize(cross(iN, B));
    float w[4];
    component_weights(sd, w);
    pdf = 0;
    f3 value = mk3(0);
    if (w[0] + w[1] > 0) {
        const f3 m = normalize(wiw + wow);
        if (w[0] > 0) pdf += w[0] * evaluate_diffuse(sd, iN, wow, wiw, m, value);
        if (w[1] > 0) pdf += w[1] * evaluate_sheen(sd, wiw, m, value);     // sic: overwrites the diffuse value (disney.cuh:373)
    }
    if (w[2] + w[3] > 0) {
        const f3 wol = w2t(wow, iN, T, B);
        const f3 wil = w2t(wiw, iN, T, B);
        const f3 m = normalize(wol + wil);
        if (w[2] > 0) {
            float ax, ay;
            alpha_from_roughness(rough, aniso, ax, ay);
            f3 contrib = mk3(0);
            const float p = evaluate_mf(MDF_GGX, sd, ax, ay, wol, wil, m, contrib);
            if (p > 0) { pdf += w[2] * p; value += contrib; }
        }
        if (w[3] > 0) {
            const float alpha = clearcoat_roughness(sd);
            f3 contrib = mk3(0);
            const float p = evaluate_mf(MDF_GTR1, sd, alpha, alpha, wol, wil, m, contrib);
            if (p > 0) { pdf += w[3] * p; value += contrib; }
        }
    }
    pdf = (pdf * (1.f - transmission));
    pdf += (stPDF * transmission);
    return (stBSDF * transmission) + (value * (1.f - transmission));
}

}  // namespace orc
