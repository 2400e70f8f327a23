// lm_hooks.h — known-answer hooks: kernels (and their launch wrappers) that run the product's DEVICE functions and whole kernels on rows handed over by the
// suite (C ABI: lumen_mi_test_*, include/lumen_mi.h; host side: kat.cpp, renderer.cpp).  They are what makes kernel-level parity against the reference's own rows
// possible (tests/golden/ref_kat*.npz) — and they are TEST SURFACE: built with LUMEN_MI_TEST_HOOKS=1 (the default, `make`: the suite needs them), left out of the
// library entirely with `make HOOKS=0` (no lumen_mi_test_* symbol, no hook kernel, the table entries are null).  Included by kernels.hip only, in two places
// (LM_HOOKS_PART 1: kernels, behind the device code they call; 2: launch wrappers, in front of the kernel table).
#if LM_HOOKS_PART == 1
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_test_bsdf)(uint32_t n, int mode, const float* __restrict__ mat, const float* __restrict__ N, const float* __restrict__ T,
               const float* __restrict__ wo, const float* __restrict__ aux, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const LmMaterial sd = lm_material_from23(mat + 23u * i);
    const lf3 n3 = v3(N[3*i], N[3*i+1], N[3*i+2]), t3 = v3(T[3*i], T[3*i+1], T[3*i+2]), wo3 = v3(wo[3*i], wo[3*i+1], wo[3*i+2]);
    if (mode == 0) {
        float pdf = 0.f;
        const lf3 b = lm_evaluate_bsdf(sd, n3, t3, wo3, v3(aux[3*i], aux[3*i+1], aux[3*i+2]), pdf);
        out[8*i] = b.x; out[8*i+1] = b.y; out[8*i+2] = b.z; out[8*i+3] = pdf; out[8*i+4] = 0; out[8*i+5] = 0; out[8*i+6] = 0; out[8*i+7] = 0;
    } else if (mode == 2) {
        // the contracted evaluation the fast ReSTIR mode runs (lm_quick_setup + lm_quick_eval): out[4] = 1 where it applies
        // (lm_quick_contracts and a light above the horizon, what lm_score_quick hands it), else the row is left zero
        const lf3 wi3 = v3(aux[3*i], aux[3*i+1], aux[3*i+2]);
        const float cin = dot3(wi3, n3);
        float* o = out + 8u * i;
        for (int k = 0; k < 8; k++) o[k] = 0.f;
        if (lm_quick_contracts(sd) && cin > 0.f) {
            LmQuick Q; lm_quick_setup(sd, n3, wo3, Q);
            float pdf = 0.f;
            const lf3 b = lm_quick_eval(Q, wi3, cin, pdf);
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = pdf; o[4] = 1.f;
        }
    } else {
        float pdf = 0.f; bool spec = false; lf3 wi = v3(0.f);
        const lf3 b = lm_sample_bsdf(sd, n3, n3, t3, wo3, 1.f, aux[3*i], aux[3*i+1], aux[3*i+2], wi, pdf, spec);
        out[8*i] = b.x; out[8*i+1] = b.y; out[8*i+2] = b.z; out[8*i+3] = wi.x; out[8*i+4] = wi.y; out[8*i+5] = wi.z; out[8*i+6] = pdf; out[8*i+7] = spec ? 1.f : 0.f;
    }
}
// known-answer hook for the reservoir / CDF / output-quantisation functions (tests/golden/ref_kat.npz rows resv, cdfq, color):
//   mode 0  n sequences of 8 Reservoir updates: a = weights, b = solid-angle pdfs, c = seeds (n * 8 each);
//           out[33 * i + 4 * k ..] = (weightSum, sampleCount, id of the held sample, taken) after update k, out[33 * i + 32] = weight
//   mode 1  a = prefix sums of n weights, b = m query values; out[2 * j] = index bits, out[2 * j + 1] = pdf
//   mode 2  a = n linear values; out[j] = sRGB8 level
//   mode 3 / 5  Resample (ReSTIRKernels.cu:1259-1325), exact / fast policy: a = n surfaces (35 floats: position normal tangent incoming mat23),
//           b = n light samples (14: radiance normal position area contribution solidAnglePdf); out[5 * i ..] = contribution, pdf, applies
//   mode 4 / 6  CombineBiased of two reservoirs (:1200-1257), exact / fast: a = n surfaces, b = 2n reservoirs (17: weightSum sampleCount
//           weight sample(14)), c = n seeds; out[18 * i ..] = reservoir(17), applies
//   "applies" = 1, except in the fast modes for surfaces the contracted evaluation does not cover (those take the exact launch: LM_RARE)
template <class A> __device__ void lm_test_resample_combine(bool combine, uint32_t i, const float* __restrict__ a, const float* __restrict__ b, const uint32_t* __restrict__ c, float* __restrict__ out)
{
    const float* sv = a + 35u * i;
    LmSurface px;
    px.position = v3(sv[0], sv[1], sv[2]); px.normal = v3(sv[3], sv[4], sv[5]); px.tangent = v3(sv[6], sv[7], sv[8]); px.incoming = v3(sv[9], sv[10], sv[11]);
    px.transport = v3(1.f); px.t = 1.f; px.flags = 0u;
    px.mat = lm_material_from23(sv + 12);
    const bool applies = !A::contracted || lm_quick_contracts(px.mat);
    auto sample = [](const float* v) { LmSample s; s.p.radiance = v3(v[0], v[1], v[2]); s.p.normal = v3(v[3], v[4], v[5]); s.p.position = v3(v[6], v[7], v[8]); s.p.area = v[9];
                                        s.contribution = v3(v[10], v[11], v[12]); s.pdf = v[13]; return s; };
    LmTarget t; lm_target_setup<A>(px, t);
    if (!combine) {
        float* o = out + 5u * i;
        LmSample rs; lm_resample<A>(sample(b + 14u * i), t, rs);
        o[0] = rs.contribution.x; o[1] = rs.contribution.y; o[2] = rs.contribution.z; o[3] = rs.pdf; o[4] = applies ? 1.f : 0.f;
    } else {
        LmReservoir r[2], dst;
        for (uint32_t k = 0; k < 2u; k++) { const float* v = b + 17u * (2u * i + k); r[k].weightSum = v[0]; r[k].count = (long long)v[1]; r[k].weight = v[2]; r[k].s = sample(v + 3); }
        lm_combine2<A>(dst, r[0], r[1], t, c[i]);
        float* o = out + 18u * i;
        o[0] = dst.weightSum; o[1] = (float)dst.count; o[2] = dst.weight;
        o[3] = dst.s.p.radiance.x; o[4] = dst.s.p.radiance.y; o[5] = dst.s.p.radiance.z; o[6] = dst.s.p.normal.x; o[7] = dst.s.p.normal.y; o[8] = dst.s.p.normal.z;
        o[9] = dst.s.p.position.x; o[10] = dst.s.p.position.y; o[11] = dst.s.p.position.z; o[12] = dst.s.p.area;
        o[13] = dst.s.contribution.x; o[14] = dst.s.contribution.y; o[15] = dst.s.contribution.z; o[16] = dst.s.pdf; o[17] = applies ? 1.f : 0.f;
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_test_restir)(int mode, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, const uint32_t* __restrict__ c, uint32_t m, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (mode >= 3) {
        if (i >= n) return;
        if (mode == 3 || mode == 4) lm_test_resample_combine<LmExact>(mode == 4, i, a, b, c, out);
        else lm_test_resample_combine<LmFast>(mode == 6, i, a, b, c, out);
        return;
    }
    if (mode == 0) {
        if (i >= n) return;
        LmReservoir r; lm_res_fresh(r);
        for (uint32_t k = 0; k < 8u; k++) {
            LmSample s; lm_sample_zero(s); s.p.area = (float)(k + 1u); s.pdf = b[8u * i + k];
            const bool took = lm_res_update<LmExact>(r, s, a[8u * i + k], c[8u * i + k]);
            float* o = out + 33u * i + 4u * k;
            o[0] = r.weightSum; o[1] = (float)r.count; o[2] = r.s.p.area; o[3] = took ? 1.f : 0.f;
        }
        lm_res_update_weight<LmExact>(r);
        out[33u * i + 32u] = r.weight;
    } else if (mode == 1) {
        if (i >= m) return;
        LmScene sc{};
        sc.cdf = a; sc.numLights = n; sc.cdfSum = a[n - 1u];
        uint32_t idx; float pdf;
        lm_cdf_get(sc, b[i], idx, pdf);
        out[2u * i] = u2f(idx); out[2u * i + 1u] = pdf;
    } else {
        if (i >= n) return;
        out[i] = (float)lm_srgb8(a[i]);
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// Known-answer hooks for whole KERNELS (tests/golden/ref_kat5.npz: what the reference's own __global__ bodies computed, oracle/ref_kat/gen_kat5.cpp).
// The host side (kat.cpp) lays synthetic surfaces / reservoirs out with the product's own store functions below, launches the product's kernels
// through the kernel table exactly as frame.cpp does, and reads the buffers back through the product's load functions.
// Rows are 32-bit words: floats by bit pattern; surface(40) = flags t position normal tangent incoming transport mat23; reservoir(17) = weightSum
// sampleCount weight radiance normal position area contribution solidAnglePdf.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ LmSurface lm_kat_surface(const uint32_t* __restrict__ w)
{
    LmSurface s;
    s.flags = w[0]; s.t = u2f(w[1]);
    s.position = v3(u2f(w[2]), u2f(w[3]), u2f(w[4])); s.normal = v3(u2f(w[5]), u2f(w[6]), u2f(w[7])); s.tangent = v3(u2f(w[8]), u2f(w[9]), u2f(w[10]));
    s.incoming = v3(u2f(w[11]), u2f(w[12]), u2f(w[13])); s.transport = v3(u2f(w[14]), u2f(w[15]), u2f(w[16]));
    float m[23];
    for (int i = 0; i < 23; i++) m[i] = u2f(w[17 + i]);
    if (s.flags == 0u) s.mat = lm_material_from23(m);
    else { s.mat.color = make_float4(m[0], m[1], m[2], m[3]); s.mat.tint = make_float4(0.f, 0.f, 0.f, 0.f); s.mat.transmittance = make_float4(0.f, 0.f, 0.f, 0.f); s.mat.p0 = s.mat.p1 = s.mat.p2 = 0u; }
    return s;
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_pack_surfaces)(const uint32_t* __restrict__ rows40, uint32_t n, float4* gbuf, float4* probe)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const LmSurface s = lm_kat_surface(rows40 + 40u * i);
    lm_gbuf_store(gbuf, probe, i, s);
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_reservoirs)(uint32_t* rows17, uint32_t n, float4* hot, float4* contrib, int unpack)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t* w = rows17 + 17u * i;
    LmReservoir r;
    if (unpack) {
        lm_res_load(hot, contrib, i, r);
        w[0] = f2u(r.weightSum); w[1] = (uint32_t)r.count; w[2] = f2u(r.weight);
        w[3] = f2u(r.s.p.radiance.x); w[4] = f2u(r.s.p.radiance.y); w[5] = f2u(r.s.p.radiance.z); w[6] = f2u(r.s.p.normal.x); w[7] = f2u(r.s.p.normal.y); w[8] = f2u(r.s.p.normal.z);
        w[9] = f2u(r.s.p.position.x); w[10] = f2u(r.s.p.position.y); w[11] = f2u(r.s.p.position.z); w[12] = f2u(r.s.p.area);
        w[13] = f2u(r.s.contribution.x); w[14] = f2u(r.s.contribution.y); w[15] = f2u(r.s.contribution.z); w[16] = f2u(r.s.pdf);
    } else {
        r.weightSum = u2f(w[0]); r.count = (long long)w[1]; r.weight = u2f(w[2]);
        r.s.p.radiance = v3(u2f(w[3]), u2f(w[4]), u2f(w[5])); r.s.p.normal = v3(u2f(w[6]), u2f(w[7]), u2f(w[8])); r.s.p.position = v3(u2f(w[9]), u2f(w[10]), u2f(w[11])); r.s.p.area = u2f(w[12]);
        r.s.contribution = v3(u2f(w[13]), u2f(w[14]), u2f(w[15])); r.s.pdf = u2f(w[16]);
        lm_res_store(hot, contrib, i, r);
    }
}
// the visibility queue of pass `pass` resolved from a per-pixel mask instead of the tracer (the OptiX programs are closed; the rows carry a mask)
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_resolve)(LmFrame fr, int rc, const uint32_t* __restrict__ countPtr, const uint8_t* __restrict__ occluded, int pass)
{
    rc = lm_res_idx(fr, rc);
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= *countPtr) return;
    const float4* __restrict__ qD = pass ? fr.vis2D : fr.visD;
    const uint32_t li = f2u(qD[i].w);
    lm_vis_resolve(fr, rc, fr.res[rc], li, occluded[li] != 0, pass);
}
// ExtractSurfaceDataGpu (GPUExtractSurfaceData.cu:8-228) as every wave kernel runs it: lm_extract on (hit record, ray) rows against the renderer's current scene.
// hits9: entry prim baryU baryV (binary16 bits) t px py - -; rays9: origin direction contribution.  out35: flags t position normal geomNormal(0: not kept by this
// build, the reference never reads it) tangent incoming transport color4 tint4 transmittance4 params3
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_extract)(LmScene sc, uint32_t n, const uint32_t* __restrict__ hits9, const uint32_t* __restrict__ rays9, uint32_t* __restrict__ out35)
{
    __shared__ float s_lut[256];
    __shared__ uint4 s_tab[LM_TABLE_QUADS];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const LmTables tab = lm_stage_tables(s_tab, sc);
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t* h = hits9 + 9u * i; const uint32_t* r = rays9 + 9u * i;
    const uint4 rec = make_uint4(h[0], h[1], (h[2] & 0xffffu) | (h[3] << 16), h[4]);
    LmSurface s;
    lm_extract(sc, lut, tab, rec, v3(u2f(r[0]), u2f(r[1]), u2f(r[2])), v3(u2f(r[3]), u2f(r[4]), u2f(r[5])), v3(u2f(r[6]), u2f(r[7]), u2f(r[8])), s);
    uint32_t* o = out35 + 35u * i;
    o[0] = s.flags; o[1] = f2u(s.t);
    const lf3 v[6] = {s.position, s.normal, v3(0.f), s.tangent, s.incoming, s.transport};
    for (int k = 0; k < 6; k++) { o[2 + 3 * k] = f2u(v[k].x); o[3 + 3 * k] = f2u(v[k].y); o[4 + 3 * k] = f2u(v[k].z); }
    const float4 q[3] = {s.mat.color, s.mat.tint, s.mat.transmittance};
    for (int k = 0; k < 3; k++) { o[20 + 4 * k] = f2u(q[k].x); o[21 + 4 * k] = f2u(q[k].y); o[22 + 4 * k] = f2u(q[k].z); o[23 + 4 * k] = f2u(q[k].w); }
    o[32] = s.mat.p0; o[33] = s.mat.p1; o[34] = s.mat.p2;
}
// the texture fetch every extraction runs (lm_tex2D: tex2D<float4> on a PTTexture object, PTTexture.cpp:35-74) on given coordinates of one texture
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_tex2d)(LmScene sc, uint32_t n, int id, const float2* __restrict__ uv, float4* __restrict__ out)
{
    __shared__ float s_lut[256];
    const lm_lds_float* lut = lm_stage_lut(s_lut, sc);
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i < n) out[i] = lm_tex2D(sc, lut, id, uv[i].x, uv[i].y);
}
// ShadeDirect / ShadeIndirect (GPUShadeDirect.cu:42-153, GPUShadeIndirect.cu:7-146) as the wave kernels call them; rows (x, y, seed, surface(40))
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_kat_shade)(LmScene sc, uint32_t n, uint32_t W, const uint32_t* __restrict__ rows43, int fast, uint32_t* __restrict__ direct12, uint32_t* __restrict__ indirect10)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = rows43 + 43u * i;
    const uint32_t gi = w[1] * W + w[0], seed = w[2];
    const LmSurface s = lm_kat_surface(w + 3);
    if (direct12) {
        uint32_t* o = direct12 + 12u * i;
        lf3 dir = v3(0.f), rad = v3(0.f); float tmax = 0.f;
        const bool emit = fast ? lm_shade_direct<LmFast>(sc, s, gi, seed, dir, tmax, rad) : lm_shade_direct<LmExact>(sc, s, gi, seed, dir, tmax, rad);
        for (int k = 0; k < 12; k++) o[k] = 0u;
        if (emit) {
            o[0] = 1u; o[1] = f2u(s.position.x); o[2] = f2u(s.position.y); o[3] = f2u(s.position.z); o[4] = f2u(dir.x); o[5] = f2u(dir.y); o[6] = f2u(dir.z); o[7] = f2u(tmax);
            o[8] = f2u(rad.x); o[9] = f2u(rad.y); o[10] = f2u(rad.z); o[11] = 1u;                     // the wave kernels add NEE light to INDIRECT (LightChannel::INDIRECT = 1)
        }
    }
    if (indirect10) {
        uint32_t* o = indirect10 + 10u * i;
        lf3 org = v3(0.f), dir = v3(0.f), con = v3(0.f);
        const bool emit = lm_shade_indirect(s, gi, seed, org, dir, con);
        for (int k = 0; k < 10; k++) o[k] = 0u;
        if (emit) { o[0] = 1u; o[1] = f2u(org.x); o[2] = f2u(org.y); o[3] = f2u(org.z); o[4] = f2u(dir.x); o[5] = f2u(dir.y); o[6] = f2u(dir.z); o[7] = f2u(con.x); o[8] = f2u(con.y); o[9] = f2u(con.z); }
    }
}
extern "C" __global__ void __launch_bounds__(LM_BLOCK)
KN(lm_k_test_math)(uint32_t n, int fn, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * LM_BLOCK + threadIdx.x;
    if (i >= n) return;
    float s, c;
    switch (fn) {
    case 0: lm_sincosf(x[i], &s, &c); out[i] = s; break;
    case 1: lm_sincosf(x[i], &s, &c); out[i] = c; break;
    case 2: out[i] = lm_logf(x[i]); break;
    case 3: out[i] = lm_expf(x[i]); break;
    case 4: out[i] = lm_powf(x[i], y[i]); break;
    case 5: out[i] = lm_halton(f2u(x[i]), f2u(y[i])); break;
    case 6: out[i] = u2f(lm_wang_hash(f2u(x[i]))); break;
    case 7: out[i] = u2f(lm_f32_to_f16(x[i])); break;
    default: out[i] = lm_f16_to_f32(f2u(x[i])); break;
    }
}
#else
static void l_test_bsdf(hipStream_t s, uint32_t n, int mode, const float* mat, const float* N, const float* T, const float* wo, const float* aux, float* out)
{ hipLaunchKernelGGL(KN(lm_k_test_bsdf), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), n, mode, mat, N, T, wo, aux, out); }
static void l_test_restir(hipStream_t s, int mode, uint32_t n, const float* a, const float* b, const uint32_t* c, uint32_t m, float* out)
{ const uint32_t t = mode == 1 ? m : n; hipLaunchKernelGGL(KN(lm_k_test_restir), LM_GRID((t + LM_BLOCK - 1) / LM_BLOCK), mode, n, a, b, c, m, out); }
static void l_kat_pack_surfaces(hipStream_t s, const uint32_t* rows40, uint32_t n, float4* gbuf, float4* probe) { hipLaunchKernelGGL(KN(lm_k_kat_pack_surfaces), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), rows40, n, gbuf, probe); }
static void l_kat_reservoirs(hipStream_t s, uint32_t* rows17, uint32_t n, float4* hot, float4* contrib, int unpack) { hipLaunchKernelGGL(KN(lm_k_kat_reservoirs), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), rows17, n, hot, contrib, unpack); }
static void l_kat_resolve(hipStream_t s, LmFrame fr, int rc, const uint32_t* count, const uint8_t* occluded, int pass) { hipLaunchKernelGGL(KN(lm_k_kat_resolve), LM_GRID((fr.n + LM_BLOCK - 1) / LM_BLOCK), fr, rc, count, occluded, pass); }
static void l_kat_extract(hipStream_t s, LmScene sc, uint32_t n, const uint32_t* hits9, const uint32_t* rays9, uint32_t* out35) { hipLaunchKernelGGL(KN(lm_k_kat_extract), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), sc, n, hits9, rays9, out35); }
static void l_kat_tex2d(hipStream_t s, LmScene sc, uint32_t n, int id, const float2* uv, float4* out) { hipLaunchKernelGGL(KN(lm_k_kat_tex2d), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), sc, n, id, uv, out); }
static void l_kat_shade(hipStream_t s, LmScene sc, uint32_t n, uint32_t W, const uint32_t* rows43, int fast, uint32_t* direct12, uint32_t* indirect10)
{ hipLaunchKernelGGL(KN(lm_k_kat_shade), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), sc, n, W, rows43, fast, direct12, indirect10); }
static void l_test_math(hipStream_t s, uint32_t n, int fn, const float* x, const float* y, float* out) { hipLaunchKernelGGL(KN(lm_k_test_math), LM_GRID((n + LM_BLOCK - 1) / LM_BLOCK), n, fn, x, y, out); }
#endif
#undef LM_HOOKS_PART
