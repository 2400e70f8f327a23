// lm_restir.h — device code: ReSTIR DI reservoirs (storage, update, resampling, biased combination).  Included by kernels.hip only.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// ReSTIR DI — reference ReSTIRData.h:115-178, ReSTIRKernels.cu, Framework/ReSTIR.cpp:65-233
// reservoir storage: a 64-byte "hot" record per pixel (what reuse passes gather from other pixels):
//   0 (weightSum, weight, sampleCount bits, solidAnglePdf)   1 (radiance, area)   2 (normal, 0)   3 (position, 0)
// plus a separate plane with the unshadowed contribution (only ever read for the pixel being shaded).
// ---------------------------------------------------------------------------------------------------------------------
struct LmSample { lf3 radiance, normal, position, contribution; float area, pdf; };
struct LmReservoir { float weightSum, weight; long long count; LmSample s; };

__device__ __forceinline__ void lm_sample_zero(LmSample& s) { s.radiance = v3(0.f); s.normal = v3(0.f); s.position = v3(0.f); s.contribution = v3(0.f); s.area = 0.f; s.pdf = 0.f; }
__device__ __forceinline__ void lm_res_fresh(LmReservoir& r) { r.weightSum = 0.f; r.weight = 0.f; r.count = 0; lm_sample_zero(r.s); }
__device__ __forceinline__ void lm_res_unpack(const float4& a, const float4& p1, const float4& p2, const float4& p3, LmReservoir& r)
{
    r.weightSum = a.x; r.weight = a.y; r.count = (long long)f2u(a.z); r.s.pdf = a.w;
    r.s.radiance = v3(p1); r.s.area = p1.w; r.s.normal = v3(p2); r.s.position = v3(p3);
}
__device__ __forceinline__ void lm_res_load(const float4* __restrict__ hot, const float4* __restrict__ contrib, uint32_t li, LmReservoir& r)
{
    const float4* h = hot + 4u * li;
    lm_res_unpack(h[0], h[1], h[2], h[3], r);
    r.s.contribution = v3(contrib[li]);
}
__device__ __forceinline__ void lm_res_store(float4* __restrict__ hot, float4* __restrict__ contrib, uint32_t li, const LmReservoir& r)
{
    float4* h = hot + 4u * li;
    h[0] = make_float4(r.weightSum, r.weight, u2f((uint32_t)r.count), r.s.pdf);
    h[1] = v4(r.s.radiance, r.s.area);
    h[2] = v4(r.s.normal, 0.f);
    h[3] = v4(r.s.position, 0.f);
    contrib[li] = v4(r.s.contribution, 0.f);
}
__device__ __forceinline__ void lm_res_update(LmReservoir& r, const LmSample& s, float w, uint32_t seed /* by value: reference quirk */)
{
    r.weightSum += w;
    ++r.count;
    const float rnd = lm_random_float(seed);
    if (rnd <= (w / r.weightSum)) r.s = s;
}
__device__ __forceinline__ void lm_res_update_weight(LmReservoir& r)
{
    if (r.count == 0 || r.weightSum <= 0.f) { r.weight = 0; return; }
    r.weight = (1.f / fmaxf(r.s.pdf, 1.1920928955078125e-7f)) * ((1.f / (float)r.count) * r.weightSum);
}
// Resample — ReSTIRKernels.cu:1259-1325
__device__ void lm_resample(const LmSample& in, const LmSurface& px, LmSample& out)
{
    out = in;
    lf3 toLight = in.position - px.position;
    const float lDistance = length3(toLight);
    toLight = toLight / lDistance;
    const float cosIn = fmaxf(dot3(toLight, px.normal), 0.f);
    const float cosOut = fmaxf(dot3(in.normal, -toLight), 0.f);
    if (cosIn <= 0 || cosOut <= 0 || lDistance <= 0.01f) { out.pdf = 0; return; }
    const float solidAngle = (cosOut * in.area) / (lDistance * lDistance);
    float pdf = 0.f;
    const lf3 bsdf = lm_evaluate_bsdf(px.mat, px.normal, px.tangent, -px.incoming, toLight, pdf);
    const float added = pdf + bsdf.x + bsdf.y + bsdf.z;
    if (pdf <= LM_EPSILON || added != added || fabsf(added) == u2f(0x7f800000u)) { out.contribution = v3(0.f); out.pdf = 0; return; }
    const lf3 contribution = (bsdf / pdf) * solidAngle * cosIn * out.radiance;
    out.contribution = contribution;
    out.pdf = (contribution.x + contribution.y + contribution.z) / 3.f;
}
// CombineBiased for two reservoirs — ReSTIRKernels.cu:1200-1257
__device__ void lm_combine2(LmReservoir& dst, const LmReservoir& a, const LmReservoir& b, const LmSurface& px, uint32_t seed)
{
    LmReservoir out; lm_res_fresh(out);
    LmSample rs;
    lm_resample(a.s, px, rs);
    lm_res_update(out, rs, (float)a.count * a.weight * rs.pdf, seed);
    lm_resample(b.s, px, rs);
    lm_res_update(out, rs, (float)b.count * b.weight * rs.pdf, seed);
    out.count = a.count + b.count;
    lm_res_update_weight(out);
    dst = out;
}
