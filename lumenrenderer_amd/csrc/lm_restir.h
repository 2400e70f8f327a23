// lm_restir.h — device code: ReSTIR DI reservoirs (storage, streaming update, target function, biased merge).
// Included by kernels.hip only.
//
// Behaviour: reference ReSTIRData.h:115-178 (Reservoir), ReSTIRKernels.cu:1259-1325 (Resample), :1200-1257 (CombineBiased),
// Framework/ReSTIR.cpp:65-233.  Shape: a resampling loop scores MANY light points against ONE receiving surface, so the receiver is
// prepared once (LmTarget = position + the BSDF lobes of lm_bsdf.h set up for its view direction) and a candidate only pays for what
// depends on the light point.  Every function takes the arithmetic policy of lm_bsdf.h (LmExact: the bit-exact contract; LmFast:
// hardware reciprocal / square root for the target function and the resampling weights).
//
// reservoir storage: a 64-byte "hot" record per pixel, ordered by who reads it — a reuse pass gathers only quads 1..3 from OTHER pixels
// (three loads per neighbour instead of four), quad 0 is only ever read for the pixel itself:
//   0 (weightSum, solidAnglePdf, 0, 0)   1 (weight, sampleCount bits, normal.x, normal.y)   2 (radiance, area)   3 (position, normal.z)
// plus a separate plane with the unshadowed contribution (only ever read for the pixel being shaded).
// Spare words q0[2], q0[3] (lazy reuse, kernels.hip lm_vis_resolve / lm_restir_spatial_body): when a frame's history passes are deferred, the second visibility pass
// parks the weight it zeroes as (q0[2] = weight, q0[3] = 1.f) for the first spatial pass that runs later.  INVARIANT: nothing else ever reads these two words —
// lm_res_unpack ignores them, lm_k_history_copy exports the 80-byte reservoir through lm_res_load — and lm_res_store rewrites them to (0, 0).  A parked flag that
// the deferred combine does not rewrite (passes dropped: the swap chain had not turned) therefore stays in the record harmlessly until the next store to that pixel.
#pragma once

struct LmLightPoint { lf3 position, normal, radiance; float area; };       // a point on an emissive triangle, as a reservoir remembers it
struct LmSample { LmLightPoint p; lf3 contribution; float pdf; };            // + what it is worth at the surface it was last scored for
struct LmReservoir { float weightSum, weight; long long count; LmSample s; };

__device__ __forceinline__ void lm_sample_zero(LmSample& s) { s.p.radiance = v3(0.f); s.p.normal = v3(0.f); s.p.position = v3(0.f); s.p.area = 0.f; s.contribution = v3(0.f); s.pdf = 0.f; }
__device__ __forceinline__ void lm_res_fresh(LmReservoir& r) { r.weightSum = 0.f; r.weight = 0.f; r.count = 0; lm_sample_zero(r.s); }
__device__ __forceinline__ LmLightPoint lm_point_unpack(const float4& q1, const float4& q2, const float4& q3)
{
    LmLightPoint p; p.radiance = v3(q2); p.area = q2.w; p.normal = v3(q1.z, q1.w, q3.w); p.position = v3(q3); return p;
}
__device__ __forceinline__ float lm_hot_weight(const float4& q1) { return q1.x; }
__device__ __forceinline__ long long lm_hot_count(const float4& q1) { return (long long)f2u(q1.y); }
__device__ __forceinline__ void lm_res_unpack(const float4& q0, const float4& q1, const float4& q2, const float4& q3, LmReservoir& r)
{
    r.weightSum = q0.x; r.s.pdf = q0.y; r.weight = lm_hot_weight(q1); r.count = lm_hot_count(q1);
    r.s.p = lm_point_unpack(q1, q2, q3);
}
// weight = 0 in place (an occluded or flagged pixel's reservoir keeps everything else)
__device__ __forceinline__ float* lm_hot_weight_at(float4* hot, uint32_t li) { return (float*)(hot + 4u * li + 1u); }       // the weight word of pixel li
__device__ __forceinline__ void lm_hot_zero_weight(float4* hot, uint32_t li) { *lm_hot_weight_at(hot, li) = 0.f; }
__device__ __forceinline__ void lm_res_load(const float4* __restrict__ hot, const float4* __restrict__ contrib, uint32_t li, LmReservoir& r)
{
    const float4* h = hot + 4u * li;
    lm_res_unpack(h[0], h[1], h[2], h[3], r);
    r.s.contribution = v3(contrib[li]);
}
__device__ __forceinline__ void lm_res_store(float4* __restrict__ hot, float4* __restrict__ contrib, uint32_t li, const LmReservoir& r)
{
    float4* h = hot + 4u * li;
    h[0] = make_float4(r.weightSum, r.s.pdf, 0.f, 0.f);
    h[1] = make_float4(r.weight, u2f((uint32_t)r.count), r.s.p.normal.x, r.s.p.normal.y);
    h[2] = v4(r.s.p.radiance, r.s.p.area);
    h[3] = v4(r.s.p.position, r.s.p.normal.z);
    contrib[li] = v4(r.s.contribution, 0.f);
}

// the receiving surface of a resampling loop
struct LmTarget {
    lf3 position;
    LmLobes lobes;                    // exact policy: lobes.N is the shading normal, lobes.wo the direction back along the camera path
    LmQuick quick;                    // fast policy: the contracted evaluation's own (smaller) setup; whichever part a kernel does not use is never computed
};
template <class A> __device__ __forceinline__ void lm_target_setup(const LmSurface& px, LmTarget& t)
{
    t.position = px.position;
    if constexpr (A::contracted) lm_quick_setup(px.mat, px.normal, -px.incoming, t.quick);
    else lm_lobes_setup<A>(px.mat, px.normal, px.tangent, -px.incoming, t.lobes);
}
// Which kernel instantiation scores a receiving surface (kernels.hip runs the ReSTIR passes in up to two launches):
//   LM_ALL     every surface, exact policy — the default mode, bit-identical to the oracle
//   LM_COMMON  fast mode, first launch: surfaces whose lobes the contracted evaluation covers (lm_quick_contracts)
//   LM_RARE    fast mode, second launch: the others (dielectric, clear coat, mirror-like), with the exact policy; the launch returns
//              at once when surface extraction counted none in the frame (LM_CNT_RARE)
enum { LM_ALL = 0, LM_COMMON = 1, LM_RARE = 2 };
template <int ROLE> __device__ __forceinline__ bool lm_role_takes(const LmMaterial& m)
{
    return ROLE == LM_ALL || lm_quick_contracts(m) == (ROLE == LM_COMMON);
}
// fast policy: the same target function in contracted form (2 rsq + 5 rcp + 1 sqrt per light point)
__device__ __forceinline__ void lm_score_quick(const LmLightPoint& p, const LmTarget& t, lf3& contribution, float& pdfOut)
{
#pragma clang fp contract(fast)
    const lf3 d = p.position - t.position;
    const float d2 = dot3(d, d), rinv = LmFast::rsqrt(d2);
    const lf3 toLight = d * rinv;
    const float cosIn = dot3(toLight, t.quick.N), cosOut = -dot3(p.normal, toLight);
    pdfOut = 0.f;
    if (!(cosIn > 0.f && cosOut > 0.f && d2 * rinv > 0.01f)) return;
    float pdf = 0.f;
    const lf3 bsdf = lm_quick_eval(t.quick, toLight, cosIn, pdf);
    const float added = pdf + bsdf.x + bsdf.y + bsdf.z;
    if (pdf <= LM_EPSILON || added != added || fabsf(added) == u2f(0x7f800000u)) { contribution = v3(0.f); return; }
    contribution = bsdf * (cosOut * p.area * rinv * rinv * cosIn * LmFast::rcp(pdf)) * p.radiance;
    pdfOut = (contribution.x + contribution.y + contribution.z) * (1.0f / 3.0f);
}

// Target function: unshadowed contribution of the light point at the receiver; its mean is the resampling pdf (0 = unusable).
template <class A> __device__ __forceinline__ void lm_score(const LmLightPoint& p, const LmTarget& t, lf3& contribution, float& pdfOut)
{
    if constexpr (A::contracted) { lm_score_quick(p, t, contribution, pdfOut); return; }
    lf3 toLight = p.position - t.position;
    const float dist = A::sqrt(dot3(toLight, toLight));
    toLight = lm_scale_inv<A>(toLight, dist);
    const float cosIn = fmaxf(dot3(toLight, t.lobes.N), 0.f);
    const float cosOut = fmaxf(dot3(p.normal, -toLight), 0.f);
    pdfOut = 0.f;
    if (cosIn <= 0 || cosOut <= 0 || dist <= 0.01f) return;            // contribution keeps what the caller put there (reference: out = in)
    const float solidAngle = A::div(cosOut * p.area, dist * dist);
    float pdf = 0.f;
    const lf3 bsdf = lm_lobes_eval<A>(t.lobes, toLight, pdf);
    const float added = pdf + bsdf.x + bsdf.y + bsdf.z;
    if (pdf <= LM_EPSILON || added != added || fabsf(added) == u2f(0x7f800000u)) { contribution = v3(0.f); return; }
    contribution = lm_scale_inv<A>(bsdf, pdf) * solidAngle * cosIn * p.radiance;
    pdfOut = A::div(contribution.x + contribution.y + contribution.z, 3.f);
}
// Resample (ReSTIRKernels.cu:1259-1325): the sample re-scored at another receiver; the stale contribution survives the early-outs
template <class A> __device__ __forceinline__ void lm_resample(const LmSample& in, const LmTarget& t, LmSample& out)
{
    out.p = in.p;
    out.contribution = in.contribution;
    lm_score<A>(in.p, t, out.contribution, out.pdf);
}

// streaming weighted reservoir update (Reservoir::Update: the seed arrives BY VALUE, so every update of one merge draws the same number)
template <class A> __device__ __forceinline__ bool lm_res_update_decide(LmReservoir& r, float w, uint32_t seed)
{
    r.weightSum += w;
    ++r.count;
    const float rnd = lm_random_float(seed);
    if constexpr (A::contracted) return w > 0.f && rnd * r.weightSum <= w;        // rnd <= w / weightSum without the division (weightSum >= w > 0)
    else return rnd <= A::div(w, r.weightSum);
}
template <class A> __device__ __forceinline__ bool lm_res_update(LmReservoir& r, const LmSample& s, float w, uint32_t seed)
{
    const bool take = lm_res_update_decide<A>(r, w, seed);
    if (take) r.s = s;
    return take;
}
template <class A> __device__ __forceinline__ void lm_res_update_weight(LmReservoir& r)
{
    if (r.count == 0 || r.weightSum <= 0.f) { r.weight = 0; return; }
    if constexpr (A::contracted) r.weight = r.weightSum * A::rcp(fmaxf(r.s.pdf, 1.1920928955078125e-7f) * (float)r.count);
    else r.weight = A::rcp(fmaxf(r.s.pdf, 1.1920928955078125e-7f)) * (A::rcp((float)r.count) * r.weightSum);
}
// CombineBiased for two reservoirs (ReSTIRKernels.cu:1200-1257)
template <class A> __device__ __forceinline__ void lm_combine2(LmReservoir& dst, const LmReservoir& a, const LmReservoir& b, const LmTarget& t, uint32_t seed)
{
    LmReservoir out; lm_res_fresh(out);
    LmSample rs;
    lm_resample<A>(a.s, t, rs);
    lm_res_update<A>(out, rs, (float)a.count * a.weight * rs.pdf, seed);
    lm_resample<A>(b.s, t, rs);
    lm_res_update<A>(out, rs, (float)b.count * b.weight * rs.pdf, seed);
    out.count = a.count + b.count;
    lm_res_update_weight<A>(out);
    dst = out;
}
// The same merge where reservoir `a` holds a sample that was last scored AT THIS receiver (the current reservoir of a pixel after the temporal
// pass: pick, temporal merge and temporal copy all leave the sample's pdf and contribution as lm_score gives them for this very surface
// record; the visibility pass in between only zeroes the weight).  Resample of it would recompute the stored values from the same
// operands — the evaluation is skipped, everything else is lm_combine2.
template <class A> __device__ __forceinline__ void lm_combine2_a_scored_here(LmReservoir& dst, const LmReservoir& a, const LmReservoir& b, const LmTarget& t, uint32_t seed)
{
    LmReservoir out; lm_res_fresh(out);
    lm_res_update<A>(out, a.s, (float)a.count * a.weight * a.s.pdf, seed);
    LmSample rs;
    lm_resample<A>(b.s, t, rs);
    lm_res_update<A>(out, rs, (float)b.count * b.weight * rs.pdf, seed);
    out.count = a.count + b.count;
    lm_res_update_weight<A>(out);
    dst = out;
}
// ... and where it is `b` that was scored here (the temporal merge: `a` = the previous frame's reservoir, re-evaluated; `b` = this frame's fresh one)
template <class A> __device__ __forceinline__ void lm_combine2_b_scored_here(LmReservoir& dst, const LmReservoir& a, const LmReservoir& b, const LmTarget& t, uint32_t seed)
{
    LmReservoir out; lm_res_fresh(out);
    LmSample rs;
    lm_resample<A>(a.s, t, rs);
    lm_res_update<A>(out, rs, (float)a.count * a.weight * rs.pdf, seed);
    lm_res_update<A>(out, b.s, (float)b.count * b.weight * b.s.pdf, seed);
    out.count = a.count + b.count;
    lm_res_update_weight<A>(out);
    dst = out;
}
