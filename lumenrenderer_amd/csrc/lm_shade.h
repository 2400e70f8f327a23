// lm_shade.h — device code: texture fetch, surface extraction, light CDF, next-event estimation and path continuation on one
// surface.  Included by kernels.hip only.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// texture fetch: RGBA8, bilinear, wrap, normalised coordinates, optional sRGB decode per texel before filtering (reference PTTexture.cpp:35-74).
// The filter follows the linear-filtering rule the CUDA C Programming Guide publishes for the texture unit the reference samples with (decision D6): wrap = the
// fractional part of the normalised coordinate, x = N frac(u); xB = x - 0.5, i = floor(xB), alpha = frac(xB) held in 1.8 fixed point (rounded to nearest here),
// nested fp32 lerps.  LmTexDesc::srgb bit 1 selects the rule of rounds 1-4 instead (no frac step, unquantised fp32 weights: tuning key tex_filter 1).
// ---------------------------------------------------------------------------------------------------------------------
// `lut` = the 256-entry sRGB decode table staged in LDS by the calling kernel (lm_stage_lut): three table reads per texel of an sRGB
// texture are ds_read_b32 instead of dependent global gathers
typedef __attribute__((address_space(3))) float lm_lds_float;
__device__ __forceinline__ const lm_lds_float* lm_stage_lut(float* s_lut, const LmScene& sc)
{
    s_lut[threadIdx.x] = sc.srgbLut ? sc.srgbLut[threadIdx.x] : 0.f;          // LM_BLOCK = 256 threads = 256 entries (no table: a scene without textures)
    __syncthreads();
    return (const lm_lds_float*)s_lut;
}
__device__ __forceinline__ float4 lm_texel(const LmScene& sc, const lm_lds_float* lut, const LmTexDesc& t, int x, int y)
{
    const uint32_t p = sc.texels[t.offset + (uint32_t)y * t.w + (uint32_t)x];
    const uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u, a = p >> 24;
    if (t.srgb & 1u) return make_float4(lut[r], lut[g], lut[b], (float)a / 255.0f);
    return make_float4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)a / 255.0f);
}
__device__ __forceinline__ int lm_wrapi(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
__device__ float4 lm_tex2D(const LmScene& sc, const lm_lds_float* lut, int id, float u, float v)
{
    if (id < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
    const LmTexDesc t = sc.texDesc[id];
    if (t.w == 1u && t.h == 1u) return lm_texel(sc, lut, t, 0, 0);           // lerp(a, a, w) == a exactly
    const bool cudaRule = (t.srgb & 2u) == 0u;
    const float x = (cudaRule ? u - floorf(u) : u) * (float)t.w - 0.5f, y = (cudaRule ? v - floorf(v) : v) * (float)t.h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    float ax = x - fx0, ay = y - fy0;
    if (cudaRule) { ax = floorf(ax * 256.0f + 0.5f) * (1.0f / 256.0f); ay = floorf(ay * 256.0f + 0.5f) * (1.0f / 256.0f); }      // 1.8 fixed point, round to nearest
    const int x0 = lm_wrapi((int)fx0, (int)t.w), y0 = lm_wrapi((int)fy0, (int)t.h);
    const int x1 = lm_wrapi(x0 + 1, (int)t.w), y1 = lm_wrapi(y0 + 1, (int)t.h);
    const float4 t00 = lm_texel(sc, lut, t, x0, y0), t10 = lm_texel(sc, lut, t, x1, y0), t01 = lm_texel(sc, lut, t, x0, y1), t11 = lm_texel(sc, lut, t, x1, y1);
    float4 r;
    r.x = lerpf(lerpf(t00.x, t10.x, ax), lerpf(t01.x, t11.x, ax), ay);
    r.y = lerpf(lerpf(t00.y, t10.y, ax), lerpf(t01.y, t11.y, ax), ay);
    r.z = lerpf(lerpf(t00.z, t10.z, ax), lerpf(t01.z, t11.z, ax), ay);
    r.w = lerpf(lerpf(t00.w, t10.w, ax), lerpf(t01.w, t11.w, ax), ay);
    return r;
}

// The scene's entry table (80 B per primitive instance) and material table (256 B) are read by every hit; scenes of up to LM_LDS_ENTRIES /
// LM_LDS_MATERIALS of them (the benchmark scene: 104 / 26) get both staged in LDS per block, so that extraction reads them with
// ds_read_b128 instead of ~ 20 global gathers per hit.  `on` false: larger scenes read global memory as before.
#ifndef LM_LDS_TABLES
#define LM_LDS_TABLES 1            // 0: A/B switch, every scene reads the tables from global memory
#endif
#define LM_LDS_ENTRIES 128u
#define LM_LDS_MATERIALS 32u
#define LM_TABLE_QUADS (5u * LM_LDS_ENTRIES + 16u * LM_LDS_MATERIALS)
static_assert(sizeof(LmEntry) == 80 && sizeof(LmDevMaterial) == 256, "table staging copies 5 / 16 quads per record");
static_assert(offsetof(LmEntry, vertBase) == 48 && offsetof(LmEntry, emissive) == 64, "lm_extract reads LmEntry by 16-byte quads");
static_assert(offsetof(LmDevMaterial, p) == 64 && offsetof(LmDevMaterial, tex) == 80 && offsetof(LmDevMaterial, constMask) == 112 && offsetof(LmDevMaterial, texConst) == 128,
              "lm_extract reads LmDevMaterial by 16-byte quads");
struct LmTables { const lm_lds_u4* ent; const lm_lds_u4* mat; bool on; };
__device__ __forceinline__ LmTables lm_stage_tables(uint4* s_tab, const LmScene& sc)
{
    LmTables t; t.ent = (const lm_lds_u4*)s_tab; t.mat = (const lm_lds_u4*)s_tab + 5u * LM_LDS_ENTRIES;
    t.on = LM_LDS_TABLES && sc.numEntries <= LM_LDS_ENTRIES && sc.numMaterials <= LM_LDS_MATERIALS;      // block-uniform
    if (t.on) {
        const uint4* ge = (const uint4*)sc.entries; const uint4* gm = (const uint4*)sc.materials;
        for (uint32_t i = threadIdx.x; i < 5u * sc.numEntries; i += LM_BLOCK) s_tab[i] = ge[i];
        for (uint32_t i = threadIdx.x; i < 16u * sc.numMaterials; i += LM_BLOCK) s_tab[5u * LM_LDS_ENTRIES + i] = gm[i];
    }
    __syncthreads();
    return t;
}
__device__ __forceinline__ float4 lm_as_float4(const uint4& v) { return make_float4(u2f(v.x), u2f(v.y), u2f(v.z), u2f(v.w)); }

// ---------------------------------------------------------------------------------------------------------------------
// surface extraction — reference GPUExtractSurfaceData.cu:8-228
// ---------------------------------------------------------------------------------------------------------------------
struct LmSurface {
    lf3 position, normal, tangent, incoming, transport;
    float t;
    uint32_t flags;
    LmMaterial mat;
};
struct LmVertex { lf3 pos; lf2 uv; lf3 normal; float4 tangent; };
__device__ __forceinline__ LmVertex lm_load_vertex(const float4* __restrict__ verts, uint32_t v)
{
    const float4 a = verts[3u * v], b = verts[3u * v + 1u], c = verts[3u * v + 2u];
    LmVertex r;
    r.pos = v3(a.x, a.y, a.z); r.uv.x = a.w; r.uv.y = b.x; r.normal = v3(b.y, b.z, b.w); r.tangent = c;
    return r;
}

__device__ void lm_extract(const LmScene& sc, const lm_lds_float* lut, const LmTables& tab, const uint4 hit, const lf3& ro, const lf3& rd, const lf3& rc, LmSurface& s)
{
    s.position = v3(0.f); s.normal = v3(0.f); s.tangent = v3(0.f); s.incoming = v3(0.f); s.transport = v3(0.f);
    s.t = 0.f; s.flags = 0u;
    s.mat.color = make_float4(0.f, 0.f, 0.f, 0.f); s.mat.transmittance = s.mat.color; s.mat.tint = s.mat.color;
    s.mat.p0 = s.mat.p1 = s.mat.p2 = 0u;
    const float t = u2f(hit.w);
    if (!(t > 0.f)) { s.flags = LM_SF_NON_INTERSECT; return; }
    // quad q of the hit's entry: 0..2 world matrix rows, 3 (vertBase, idxBase, material, mode), 4 override radiance + scale
    auto EQ = [&](uint32_t q) -> float4 { if (tab.on) return lm_as_float4(lm_lds_read4(tab.ent + 5u * hit.x + q)); return ((const float4*)(sc.entries + hit.x))[q]; };
    LmEntry e;
    { const float4 r0 = EQ(0), r1 = EQ(1), r2 = EQ(2), ids = EQ(3);
      e.m[0] = r0.x; e.m[1] = r0.y; e.m[2] = r0.z; e.m[3] = r0.w; e.m[4] = r1.x; e.m[5] = r1.y; e.m[6] = r1.z; e.m[7] = r1.w; e.m[8] = r2.x; e.m[9] = r2.y; e.m[10] = r2.z; e.m[11] = r2.w;
      e.vertBase = f2u(ids.x); e.idxBase = f2u(ids.y); e.material = f2u(ids.z); e.mode = f2u(ids.w); e.emissive = EQ(4); }
    // quad q of its material: 0 color, 1 emissive, 2 transmittance, 3 tint, 4 packed parameters, 5..6 texture ids, 7 constMask, 8..15 folded slots
    auto MQ = [&](uint32_t q) -> float4 { if (tab.on) return lm_as_float4(lm_lds_read4(tab.mat + 16u * e.material + q)); return ((const float4*)(sc.materials + e.material))[q]; };
    const uint32_t cm = f2u(MQ(7).x);
    struct __attribute__((packed, aligned(4))) Tri3 { uint32_t a, b, c; };                  // one 12-byte load instead of three gathers
    const Tri3 tri = *(const Tri3*)(sc.indices + e.idxBase + 3u * hit.y);
    const uint32_t i0 = tri.a, i1 = tri.b, i2 = tri.c;
    const LmVertex A = lm_load_vertex(sc.verts, e.vertBase + i0), B = lm_load_vertex(sc.verts, e.vertBase + i1), C = lm_load_vertex(sc.verts, e.vertBase + i2);
    const float U = lm_f16_to_f32(hit.z & 0xffffu), V = lm_f16_to_f32(hit.z >> 16), Wt = 1.f - (U + V);
    const float uvx = A.uv.x * Wt + B.uv.x * U + C.uv.x * V;
    const float uvy = A.uv.y * Wt + B.uv.y * U + C.uv.y * V;
    const float flip = A.tangent.w;
    // texture slot k: the folded constant when the slot's texture is a single texel or null (LmDevMaterial::constMask), else the fetch
    auto TEX = [&](int k) -> float4 {
        if ((cm >> k) & 1u) return MQ(8u + (uint32_t)k);
        const float4 ids = MQ(5u + ((uint32_t)k >> 2));
        const int id = (int)f2u((k & 3) == 0 ? ids.x : (k & 3) == 1 ? ids.y : (k & 3) == 2 ? ids.z : ids.w);
        return lm_tex2D(sc, lut, id, uvx, uvy);
    };
    const float4 normalMap = TEX(6);
    const float4 texColor = TEX(3);
    float4 emissive = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.mode == 0u) { emissive = MQ(1) * e.emissive.w; emissive = emissive * TEX(4); }
    else if (e.mode == 2u) emissive = e.emissive * e.emissive.w;

    const lf3 localNormal = normalize3(A.normal * Wt + B.normal * U + C.normal * V);
    const lf3 localTangent = normalize3(v3(A.tangent) * Wt + v3(B.tangent) * U + v3(C.tangent) * V);
    const lf3 normalWorld = normalize3(v3(lm_mul_m34(e.m, localNormal, 0.f)));
    const lf3 tangentWorld = normalize3(v3(lm_mul_m34(e.m, localTangent, 0.f)));
    const lf3 bitangentWorld = cross3(normalWorld, tangentWorld) * flip;
    lf3 nm = v3(normalMap.x, normalMap.y, normalMap.z);
    nm = nm * 2.f - 1.f;
    nm = normalize3(nm);
    nm = normalize3(v3(nm.x * tangentWorld.x + nm.y * bitangentWorld.x + nm.z * normalWorld.x,
                       nm.x * tangentWorld.y + nm.y * bitangentWorld.y + nm.z * normalWorld.y,
                       nm.x * tangentWorld.z + nm.y * bitangentWorld.z + nm.z * normalWorld.z));
    s.t = t;
    s.normal = nm;
    if (emissive.x > 0.f || emissive.y > 0.f || emissive.z > 0.f) {
        const float maximum = fmaxf(emissive.x, fmaxf(emissive.y, emissive.z));
        const float inv = 1.0f / maximum;
        s.mat.color = emissive * inv;
        s.flags = LM_SF_EMISSIVE;
        return;
    }
    if (texColor.w < 0.51f) {
        s.flags = LM_SF_ALPHA;
        s.position = ro + rd * t;
        s.incoming = rd;
        s.transport = rc;
        return;
    }
    const float4 matColor = MQ(0), matTransmittance = MQ(2), matTint = MQ(3), matP = MQ(4);
    const float eta = 1.f / matTransmittance.w;
    s.position = ro + rd * t;
    s.incoming = rd;
    s.transport = rc;
    s.tangent = tangentWorld;
    s.mat.color = matColor; s.mat.transmittance = matTransmittance; s.mat.tint = matTint;
    const uint32_t mp0 = f2u(matP.x), mp1 = f2u(matP.y), mp2 = f2u(matP.z);
    s.mat.p0 = mp0; s.mat.p1 = mp1; s.mat.p2 = mp2;
    const float4 mr = TEX(5);
    const float baseMetal = lm_unpack8(mp0, 0), baseRough = lm_unpack8(mp0, 24);
    lm_pack8(s.mat.p0, 0, mr.z * baseMetal);
    lm_pack8(s.mat.p0, 24, mr.y * baseRough);
    s.mat.color = texColor * matColor;
    const float4 cc = TEX(0);
    const float4 ccr = TEX(1);
    const float4 tr = TEX(2);
    const float4 tint = TEX(7);
    const lf3 finalTint = v3(tint.x, tint.y, tint.z) * v3(matTint);
    const float finalClearCoat = lm_unpack8(mp2, 0) * cc.x;
    const float gloss = lm_unpack8(mp2, 8) * (1.f - ccr.x);
    const float finalTransmission = lm_unpack8(mp2, 16) * tr.x;
    lm_pack8(s.mat.p2, 0, finalClearCoat);
    lm_pack8(s.mat.p2, 8, gloss);
    s.mat.tint = make_float4(finalTint.x, finalTint.y, finalTint.z, s.mat.tint.w);
    lm_pack8(s.mat.p2, 16, finalTransmission);
    s.mat.transmittance.w = eta;
}

// Depth-0 surface data ("G-buffer"): one 128-byte record per pixel = exactly one cache line, because ReSTIR gathers whole
// records of OTHER pixels (spatial / temporal reuse):  float4[8] =
//   0 (position, t)   1 (normal, flags bits)   2 (tangent, 0)   3 (incoming, 0)
//   4 color           5 (tint, luminance)      6 (transmittance, eta)   7 (p0, p1, p2 bits, 0)
// plus a separate 16-byte "reuse probe" plane (normal, flags ? -1 : t): all that the similarity tests need.
// (Moving the packed parameters into the first 64-byte half, beside the position, so that the temporal pass over an empty history touches half a
// line, was measured: no change in that pass's time, frame -0.3 % — the line is fetched whole.  profiles/r03_restir_shortcuts_ab.txt)
#define LM_GB_POSITION 0u
#define LM_GB_NORMAL 1u
#define LM_GB_TANGENT 2u
#define LM_GB_INCOMING 3u
#define LM_GB_COLOR 4u
#define LM_GB_TINT 5u
#define LM_GB_TRANSMITTANCE 6u
#define LM_GB_PARAMS 7u
__device__ __forceinline__ void lm_gbuf_store(float4* __restrict__ g, float4* __restrict__ probe, uint32_t li, const LmSurface& s)
{
    float4* r = g + 8u * li;
    r[LM_GB_POSITION] = v4(s.position, s.t);
    r[LM_GB_NORMAL] = v4(s.normal, u2f(s.flags));
    r[LM_GB_PARAMS] = make_float4(u2f(s.mat.p0), u2f(s.mat.p1), u2f(s.mat.p2), 0.f);
    r[LM_GB_INCOMING] = v4(s.incoming, 0.f);
    r[LM_GB_COLOR] = s.mat.color;
    r[LM_GB_TINT] = s.mat.tint;
    r[LM_GB_TRANSMITTANCE] = s.mat.transmittance;
    r[LM_GB_TANGENT] = v4(s.tangent, 0.f);
    probe[li] = v4(s.normal, s.flags ? -1.f : s.t);
}
__device__ __forceinline__ void lm_gbuf_load(const float4* __restrict__ g, uint32_t li, LmSurface& s)
{
    const float4* r = g + 8u * li;
    const float4 a = r[LM_GB_POSITION], b = r[LM_GB_NORMAL], c = r[LM_GB_TANGENT], d = r[LM_GB_INCOMING];
    s.position = v3(a); s.t = a.w; s.normal = v3(b); s.flags = f2u(b.w); s.tangent = v3(c); s.incoming = v3(d);
    s.mat.color = r[LM_GB_COLOR]; s.mat.tint = r[LM_GB_TINT]; s.mat.transmittance = r[LM_GB_TRANSMITTANCE];
    const float4 p = r[LM_GB_PARAMS];
    s.mat.p0 = f2u(p.x); s.mat.p1 = f2u(p.y); s.mat.p2 = f2u(p.z);
    s.transport = v3(1.f, 1.f, 1.f);
}

// per-pixel kernels: one 256-thread block = one 16x16 pixel tile of the window.  Tiles are enumerated in bands of 8 tile
// rows, column-major inside a band, and each XCD (blocks b, b+8, ... share one) gets a contiguous run of that order, so the
// tiles resident on one XCD form a compact patch and neighbour gathers (+-30 px) hit that XCD's L2.  Speed only.
template <uint32_t LOG_TS = 4>   // tile edge = 1 << LOG_TS: 16 for 256-thread blocks, 32 for 1024-thread blocks
__device__ __forceinline__ void lm_tile_origin(const LmFrame& fr, uint32_t& tx, uint32_t& ty, uint32_t b = blockIdx.x)      // the block's tile (block-uniform); b: a looping
{                                                                                                                          // kernel's virtual block (grid a multiple of 8)
    constexpr uint32_t TS = 1u << LOG_TS;
    const uint32_t tilesX = (fr.ww + TS - 1u) >> LOG_TS, tilesY = (fr.wh + TS - 1u) >> LOG_TS, T = tilesX * tilesY;
    const uint32_t q = T >> 3, r = T & 7u, xcd = b & 7u;
    const uint32_t t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
    const uint32_t band = t / (8u * tilesX), within = t - band * 8u * tilesX;
    const uint32_t bh = min(8u, tilesY - band * 8u);
    tx = within / bh; ty = band * 8u + within % bh;
}
template <uint32_t LOG_TS = 4>
__device__ __forceinline__ bool lm_tile_pixel(const LmFrame& fr, uint32_t& li, uint32_t& gi, uint32_t b = blockIdx.x)
{
    constexpr uint32_t TS = 1u << LOG_TS;
    uint32_t tx, ty;
    lm_tile_origin<LOG_TS>(fr, tx, ty, b);
    const uint32_t lx = tx * TS + (threadIdx.x & (TS - 1u)), ly = ty * TS + (threadIdx.x >> LOG_TS);
    if (lx >= fr.ww || ly >= fr.wh) return false;
    li = ly * fr.ww + lx;
    gi = (fr.y0 + ly) * fr.W + (fr.x0 + lx);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// lights / CDF — reference ReSTIRData.h:230-306 (CDF::Get, BinarySearch)
// ---------------------------------------------------------------------------------------------------------------------
template <class A = LmExact>
__device__ __forceinline__ void lm_cdf_get(const LmScene& sc, float value, uint32_t& index, float& pdf)
{
    const float required = sc.cdfSum * value;
    int first = 0, last = (int)sc.numLights - 1, center = 0;
    for (;;) {
        center = (last + first) / 2;
        const float higher = sc.cdf[center];
        const float lower = center != 0 ? sc.cdf[center - 1] : 0.f;
        if (required < lower && center - 1 >= first) { last = center - 1; continue; }
        if (required > higher && center + 1 <= last) { first = center + 1; continue; }
        index = (uint32_t)center;
        pdf = A::div(higher - lower, sc.cdfSum);
        return;
    }
}
struct LmTriLight { lf3 p0, p1, p2, normal, radiance; float area; };
__device__ __forceinline__ LmTriLight lm_load_light(const LmLight* __restrict__ lights, uint32_t i)
{
    const float4 a = lights[i].a, b = lights[i].b, c = lights[i].c, d = lights[i].d;
    LmTriLight l;
    l.p0 = v3(a.x, a.y, a.z); l.p1 = v3(a.w, b.x, b.y); l.p2 = v3(b.z, b.w, c.x);
    l.normal = v3(c.y, c.z, c.w); l.radiance = v3(d.x, d.y, d.z); l.area = d.w;
    return l;
}

// ---------------------------------------------------------------------------------------------------------------------
// NEE (reference GPUShadeDirect.cu:42-153) and continuation (GPUShadeIndirect.cu:7-146) on one surface
// ---------------------------------------------------------------------------------------------------------------------
// Arithmetic policy A (lm_bsdf.h): LmExact is the bit-exact contract.  LmFast (tuning key fast_shade) evaluates the light's CONTRIBUTION with hardware
// reciprocal / square root; what it can change beyond the last bits of a radiance value is a shadow ray within rounding of the two thresholds
// below (cosIn <= 0, bsdfPdf <= epsilon) — never which path continues: sampling and Russian roulette (lm_shade_indirect) stay exact in every mode.
template <class A = LmExact>
__device__ bool lm_shade_direct(const LmScene& sc, const LmSurface& s, uint32_t gi, uint32_t seedIn, lf3& dir, float& tmaxOut, lf3& radiance)
{
    uint32_t seed = lm_wang_hash(seedIn + gi);
    if (s.flags) return false;
    uint32_t index; float pdf;
    lm_cdf_get<A>(sc, lm_random_float(seed), index, pdf);
    const LmTriLight light = lm_load_light(sc.lights, index);
    const float u = lm_random_float(seed);
    const float v = lm_random_float(seed) * (1.f - u);
    const lf3 arm1 = light.p1 - light.p0, arm2 = light.p2 - light.p0;
    const lf3 lightCenter = light.p0 + (arm1 * u) + (arm2 * v);
    lf3 toLight = lightCenter - s.position;
    const float lDistance = A::sqrt(dot3(toLight, toLight));
    toLight = lm_scale_inv<A>(toLight, lDistance);
    const float cosIn = fmaxf(dot3(toLight, s.normal), 0.f);
    const float cosOut = fmaxf(0.f, dot3(light.normal, -toLight));
    if (cosIn <= 0.f || lDistance <= 0.01f) return false;
    const float solidAngle = A::div(cosOut * light.area, lDistance * lDistance);
    float bsdfPdf = 0.f;
    const lf3 bsdf = lm_evaluate_bsdf<A>(s.mat, s.normal, s.tangent, -s.incoming, toLight, bsdfPdf);
    if (bsdfPdf <= LM_EPSILON) return false;
    lf3 contribution = lm_scale_inv<A>(bsdf, bsdfPdf) * solidAngle * cosIn * light.radiance;
    contribution = contribution * (A::rcp(pdf) * s.transport);
    dir = toLight; tmaxOut = lDistance - 0.2f; radiance = contribution;
    return true;
}
__device__ bool lm_shade_indirect(const LmSurface& s, uint32_t gi, uint32_t seedIn, lf3& origin, lf3& dir, lf3& contributionOut)
{
    uint32_t seed = lm_wang_hash(seedIn + lm_wang_hash(gi));
    if (s.flags & LM_SF_ALPHA) { origin = s.position; dir = s.incoming; contributionOut = s.transport; return true; }
    if (s.flags) return false;
    if (fabsf(dot3(s.normal, s.incoming)) < 3.f * LM_EPSILON) return false;
    lf3 bounce = v3(0.f);
    float pdf = 0.f;
    bool specular = false;
    const float r0 = lm_random_float(seed), r1 = lm_random_float(seed), r2 = lm_random_float(seed);
    const lf3 bsdf = lm_sample_bsdf(s.mat, s.normal, s.normal, s.tangent, -s.incoming, 1.f, r0, r1, r2, bounce, pdf, specular);
    const float chk = pdf + bsdf.x + bsdf.y + bsdf.z;
    if (pdf <= LM_EPSILON || chk != chk) return false;
    const float rrWeight = specular ? 1.f : fminf(fmaxf(bsdf.x, fmaxf(bsdf.y, bsdf.z)), 1.f);
    const float rnd = lm_random_float(seed);
    if (rrWeight < rnd) return false;
    const float rrPdf = 1.f / rrWeight;
    lf3 contribution = s.transport * rrPdf;
    contribution = contribution * (bsdf * fabsf(dot3(s.normal, bounce)) * (1.f / pdf));
    origin = s.position; dir = bounce; contributionOut = contribution;
    return true;
}

// inside the owned tile grown by `margin` pixels? (window-local pixel index)
__device__ __forceinline__ bool lm_owned(const LmFrame& fr, uint32_t li, int margin)
{
    const int ly = (int)(li / fr.ww), lx = (int)(li - (uint32_t)ly * fr.ww);
    return lx >= (int)fr.tx0 - margin && lx < (int)fr.tx1 + margin && ly >= (int)fr.ty0 - margin && ly < (int)fr.ty1 + margin;
}
