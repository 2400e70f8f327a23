// renderer.cpp — the C ABI of liblumen_mi.so (include/lumen_mi.h): resource factories, scene graph edits, settings, frame
// entry points, read-back and queries.
//
// Mirrors the call surface of the reference's WaveFront::WaveFrontRenderer : LumenRenderer
// (LumenPT/src/Framework/WaveFrontRenderer.{h,cpp}); every extern "C" entry point is declared and cited in
// include/lumen_mi.h.  Scene flattening / light list: scene.cpp; frame graph: frame.cpp; shared state: renderer_state.h.
#include "renderer_state.h"

// ==============================================================================================================
// C ABI
// ==============================================================================================================
extern "C" {

const char* lumen_mi_last_error(void) { return g_lastError.c_str(); }

// Which kernel classes run from the compilation without the SLP vectoriser (kernels.hip LM_NOSLP_VARIANT): comma-separated table entries, "all" or "none".  Two lists: one
// that always applies and one that is added while the ReSTIR passes run in the EXACT arithmetic mode (the exact instantiations of temporal / spatial / combine gain without the
// pairing, the fast ones — short kernels on the frame's critical chain — lose: profiles/r06_noslp_kernels_ab.txt).  Environment LUMEN_MI_NOSLP_KERNELS / LUMEN_MI_NOSLP_KERNELS_EXACT
// override the defaults.  Results do not depend on the choice: the two compilations differ in instruction selection only (-ffp-contract=off in both; the suite runs under "all" too).
#ifndef LM_NOSLP_DEFAULT
#define LM_NOSLP_DEFAULT "pick_primary,extract0,shade_wave,merge"
#endif
#ifndef LM_NOSLP_DEFAULT_EXACT
#define LM_NOSLP_DEFAULT_EXACT "temporal,spatial,combine"
#endif
static void applyNoSlpKernels(lumen_mi_renderer* r)
{
    const LmKernelTable* a = lm_kernel_table();
    const LmKernelTable* b = lm_kernel_table_noslp();
    const char* e0 = getenv("LUMEN_MI_NOSLP_KERNELS");
    const char* e1 = getenv("LUMEN_MI_NOSLP_KERNELS_EXACT");
    std::string s = std::string(",") + (e0 ? e0 : LM_NOSLP_DEFAULT) + ",";
    if (!r->fastResample) s += std::string(e1 ? e1 : LM_NOSLP_DEFAULT_EXACT) + ",";
    const bool all = s.find(",all,") != std::string::npos;
    auto on = [&](const char* name) { return all || s.find(std::string(",") + name + ",") != std::string::npos; };
    r->Kmix = *a;
#define LM_MIX(entry) if (on(#entry)) r->Kmix.entry = b->entry;
    LM_MIX(primary) LM_MIX(trace_closest) LM_MIX(extract0) LM_MIX(shade_wave) LM_MIX(trace_shadow) LM_MIX(path_tail) LM_MIX(fill_bags) LM_MIX(pick_primary)
    LM_MIX(trace_shade) LM_MIX(temporal) LM_MIX(spatial) LM_MIX(combine) LM_MIX(clear) LM_MIX(merge) LM_MIX(trace_primary)
#undef LM_MIX
    if (!r->instrumented) r->K = &r->Kmix;
}

int lumen_mi_create(lumen_mi_renderer** out)
{
    if (!out) return fail(LUMEN_MI_ERR_INVALID, "out is NULL");
    initLut();
    *out = new lumen_mi_renderer();
    if (const char* e = getenv("LUMEN_MI_REFILL")) (*out)->refillBelow = atoi(e);
    if (const char* e = getenv("LUMEN_MI_REFILL_VIS")) (*out)->refillVisibility = atoi(e);
    if (const char* e = getenv("LUMEN_MI_REFILL_PRIMARY")) (*out)->refillPrimary = atoi(e);
    if (const char* e = getenv("LUMEN_MI_TAIL_BELOW")) (*out)->tailBelow = atoi(e);
    if (const char* e = getenv("LUMEN_MI_PICK_AHEAD")) (*out)->pickAhead = atoi(e);
    if (const char* e = getenv("LUMEN_MI_SHADOW_ON_WAVE")) (*out)->shadowOnWave = atoi(e);
    if (const char* e = getenv("LUMEN_MI_PACKET_PRIMARY")) (*out)->packetPrimary = atoi(e);
    if (const char* e = getenv("LUMEN_MI_FAST_SHADE")) (*out)->fastShade = atoi(e);
    if (const char* e = getenv("LUMEN_MI_SPATIAL_LDS")) (*out)->spatialLds = atoi(e);
    if (const char* e = getenv("LUMEN_MI_PICK_WIDE")) (*out)->pickWide = atoi(e);
    if (const char* e = getenv("LUMEN_MI_FUSE_COMBINE")) (*out)->fuseCombine = atoi(e);
    if (const char* e = getenv("LUMEN_MI_TAIL_REPACK")) (*out)->tailRepack = atoi(e);
    if (const char* e = getenv("LUMEN_MI_GPU_BUILD")) (*out)->gpuBuild = atoi(e);
    if (const char* e = getenv("LUMEN_MI_LAZY_REUSE")) (*out)->lazyReuse = std::max(-1, std::min(1, atoi(e)));      // (2 = the deliberately wrong test mode: tuning key only)
    if (const char* e = getenv("LUMEN_MI_FUSE_PRIMARY")) (*out)->fusePrimary = atoi(e);
    if (const char* e = getenv("LUMEN_MI_PACKET_VISIBILITY")) (*out)->packetVisibility = atoi(e);
    if (const char* e = getenv("LUMEN_MI_SORT_RAYS")) (*out)->sortRays = std::max(0, atoi(e));
    if (const char* e = getenv("LUMEN_MI_FAST_RESAMPLE")) (*out)->fastResample = atoi(e) != 0;
    if (const char* e = getenv("LUMEN_MI_WAVE_STREAMS")) (*out)->waveStreams = atoi(e) == 2 ? 2 : 1;
    if (const char* e = getenv("LUMEN_MI_TAIL_PAIR")) (*out)->tailPair = atoi(e);
    if (const char* e = getenv("LUMEN_MI_TAIL_GRID")) (*out)->tailGrid = std::max(1, std::min(8, atoi(e)));      // (<= 8: the stack-spill area is sized for 8 blocks per CU)
    if (const char* e = getenv("LUMEN_MI_TAIL_LANES")) { const int v = atoi(e); (*out)->tailLanes = v <= 0 ? -1 : std::min(64, v); }
    applyNoSlpKernels(*out);
    return 0;
}

int lumen_mi_init(lumen_mi_renderer* r, const lumen_mi_settings* s)
{
    if (!r || !s) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (s->render_width == 0 || s->render_height == 0 || s->render_width > 65535 || s->render_height > 65535) return fail(LUMEN_MI_ERR_INVALID, "render resolution must be in [1, 65535]");
    if (s->depth > LM_MAX_DEPTH) return fail(LUMEN_MI_ERR_INVALID, "depth must be in [1, 16] (0 = the reference's default, 5)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(LUMEN_MI_ERR_DEVICE, "no HIP device: the MI355X path needs a GPU (there is no CPU fallback)");
    if (s->device < 0 || s->device >= count) return fail(LUMEN_MI_ERR_INVALID, "device ordinal out of range");
    LM_HIP(hipSetDevice(s->device));
    hipDeviceProp_t prop;
    LM_HIP(hipGetDeviceProperties(&prop, s->device));
    r->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    r->device = s->device;
    r->settings = *s; r->pending = *s;
    if (r->settings.depth == 0) { r->settings.depth = 5; r->pending.depth = 5; }
    r->pending.output_width = r->settings.output_width = s->output_width ? s->output_width : s->render_width;
    r->pending.output_height = r->settings.output_height = s->output_height ? s->output_height : s->render_height;
    if (const char* e = getenv("LUMEN_MI_SINGLE_STREAM")) r->overlap = atoi(e) == 0;
    if (const char* e = getenv("LUMEN_MI_TRACE_BLOCKS_MAIN")) r->traceBlocksMain = std::max(0, std::min(8, atoi(e)));
    if (const char* e = getenv("LUMEN_MI_TRACE_BLOCKS_AUX")) r->traceBlocksAux = std::max(1, std::min(8, atoi(e)));
    if (const char* e = getenv("LUMEN_MI_TRACE_BLOCKS_VIS")) r->traceBlocksVis = std::max(0, std::min(8, atoi(e)));
    if (const char* e = getenv("LUMEN_MI_AUX_PRIORITY")) r->auxPriority = atoi(e);
    if (const char* e = getenv("LUMEN_MI_AUX3_PRIORITY")) r->aux3Priority = atoi(e);
    if (!r->aux) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);         // numerically lower = higher priority
        LM_HIP(hipStreamCreateWithPriority(&r->aux, hipStreamNonBlocking, r->auxPriority ? hi : lo));
        LM_HIP(hipStreamCreateWithPriority(&r->aux2, hipStreamNonBlocking, r->auxPriority ? hi : lo));
        LM_HIP(hipStreamCreateWithPriority(&r->aux3, hipStreamNonBlocking, r->aux3Priority ? hi : lo));
        LM_HIP(hipEventCreateWithFlags(&r->evPick, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evJoin2, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evVis, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evVisDone, hipEventDisableTiming));
        r->evShade.resize(LM_MAX_DEPTH + 1);
        for (auto& e : r->evShade) LM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evFront, hipEventDisableTiming));
        for (auto& e : r->evTemporal) LM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evTail, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evScene, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&r->evTop, hipEventDisableTiming));
        for (auto& e : r->evMerge) LM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) { LM_HIP(hipEventCreateWithFlags(&r->evCnt[i], hipEventDisableTiming)); LM_HIP(hipHostMalloc((void**)&r->pinnedCounters[i], LM_CNT_WORDS * sizeof(uint32_t), hipHostMallocDefault)); }
        LM_HIP(hipEventCreateWithFlags(&r->evJoin, hipEventDisableTiming));
    }
    r->initialised = true;
    return 0;
}

int lumen_mi_destroy(lumen_mi_renderer* r)
{
    if (!r) return 0;
    lumen_mi_stop_rendering(r);
    if (r->initialised) {
        (void)hipSetDevice(r->device);
        (void)hipStreamSynchronize(r->stream);
        if (r->aux) {
            for (hipStream_t s : {r->aux, r->aux2, r->aux3}) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
            for (hipEvent_t e : {r->evPick, r->evVis, r->evVisDone, r->evJoin, r->evJoin2, r->evFront, r->evTail, r->evTop, r->evScene}) (void)hipEventDestroy(e);
            for (auto& e : r->evShade) (void)hipEventDestroy(e);
            for (auto& e : r->evTemporal) (void)hipEventDestroy(e);
            for (auto& e : r->evMerge) (void)hipEventDestroy(e);
            for (int i = 0; i < 2; i++) { (void)hipEventDestroy(r->evCnt[i]); (void)hipHostFree(r->pinnedCounters[i]); r->pinnedCounters[i] = nullptr; }
        }
        for (SceneSet& S : r->sset) { if (S.upPending) (void)hipEventSynchronize(S.evUp); S.release(); }
        r->dSpill.release(); r->dVerts.release(); r->dIndices.release();
        r->dTriBox.release(); r->dNodeBox.release(); r->dRefitBounds.release();
        r->dMaterials.release(); r->dTexDesc.release(); r->dTexels.release(); r->dLut.release();
        for (auto& b : r->dRay) b.release();
        for (auto& b : r->dTailRay) b.release();
        for (auto& b : r->dSh) b.release();
        for (auto& b : r->dSh2) b.release();
        for (auto& b : r->dGbuf) b.release();
        for (auto& b : r->dProbe) b.release();
        for (auto& b : r->dRes) b.release();
        for (auto& b : r->dResC) b.release();
        for (auto& b : r->dMotion) b.release();
        for (int i = 0; i < 2; i++) { r->dDirect[i].release(); r->dIndirect[i].release(); }
        r->dReuseMask.release(); r->dHazard[0].release(); r->dHazard[1].release(); for (auto& b : r->dRareTile) b.release();
        r->dSortBins.release(); r->dExportHalf.release(); r->dTotals.release(); r->dCombined.release(); for (auto& b : r->dHits) b.release(); r->dCounters.release(); r->dSwap.release(); r->dOutput.release(); r->dBags.release();
        for (auto& e : r->evPool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
    delete r;
    return 0;
}

int lumen_mi_set_stream(lumen_mi_renderer* r, void* s) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); ApiLock lk(r); r->stream = (hipStream_t)s; return 0; }

int lumen_mi_create_texture(lumen_mi_renderer* r, const void* rgba8, uint32_t w, uint32_t h, int normalize, lumen_mi_handle* out)
{
    if (!r || !rgba8 || !out || w == 0 || h == 0) return fail(LUMEN_MI_ERR_INVALID, "bad texture arguments");
    ApiLock lk(r);
    Texture t; t.w = w; t.h = h; t.srgb = normalize != 0;          // a_Normalize selects sRGB decode (PTTexture.cpp:57-73)
    t.px.resize((size_t)w * h);
    memcpy(t.px.data(), rgba8, (size_t)w * h * 4);
    for (uint32_t p : t.px) t.minG = std::min<uint8_t>(t.minG, (uint8_t)((p >> 8) & 255u));
    r->textures.push_back(std::move(t));
    r->texturesDirty = true;
    *out = mkh(H_TEXTURE, r->textures.size() - 1);
    return 0;
}

int lumen_mi_create_default_resources(lumen_mi_renderer* r, lumen_mi_handle* white, lumen_mi_handle* normal, lumen_mi_handle* diffuse)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    // LumenRenderer::CreateDefaultResources (Lumen/src/Lumen/Renderer/LumenRenderer.cpp:50-58): three 1x1 textures, normalize = false
    const uint8_t w[4] = {255, 255, 255, 255}, n[4] = {128, 128, 255, 0}, d[4] = {255, 255, 255, 255};
    lumen_mi_handle hw, hn, hd; int rc;
    if ((rc = lumen_mi_create_texture(r, w, 1, 1, 0, &hw)) || (rc = lumen_mi_create_texture(r, n, 1, 1, 0, &hn)) || (rc = lumen_mi_create_texture(r, d, 1, 1, 0, &hd))) return rc;
    if (white) *white = hw; if (normal) *normal = hn; if (diffuse) *diffuse = hd;
    return 0;
}

static int fillMaterial(lumen_mi_renderer* r, const lumen_mi_material_data* d, Material& m)
{
    if (!(d->roughness_factor > 0.f)) return fail(LUMEN_MI_ERR_INVALID, "roughness factor must be > 0 (WaveFrontRenderer.cpp:1283)");
    auto tex = [&](lumen_mi_handle h, int& id) -> bool { if (h == 0) { id = -1; return true; } size_t i; if (!unh(h, H_TEXTURE, r->textures.size(), i)) return false; id = (int)i; return true; };
    int tDiff, tNorm, tMR, tEm, tTr, tCC, tCCR, tTint;
    if (!tex(d->diffuse_texture, tDiff) || !tex(d->normal_map, tNorm) || !tex(d->metallic_roughness_texture, tMR) || !tex(d->emissive_texture, tEm) ||
        !tex(d->transmission_texture, tTr) || !tex(d->clearcoat_texture, tCC) || !tex(d->clearcoat_roughness_texture, tCCR) || !tex(d->tint_texture, tTint))
        return fail(LUMEN_MI_ERR_INVALID, "bad texture handle in material");
    // the reference asserts that all eight textures are present (WaveFrontRenderer.cpp:1273-1280)
    if (tDiff < 0 || tNorm < 0 || tMR < 0 || tEm < 0 || tTr < 0 || tCC < 0 || tCCR < 0 || tTint < 0) return fail(LUMEN_MI_ERR_INVALID, "all eight material textures are required (use the default textures)");
    memset(&m, 0, sizeof m);
    LmDevMaterial& v = m.dev;
    // PTMaterial(): MaterialData(0), roughness 1 (PTMaterial.cpp:10-19), then the setters in the order of CreateMaterial
    pack8(v.p[0], 24, 1.f);
    v.color = make_float4(d->diffuse_color[0], d->diffuse_color[1], d->diffuse_color[2], d->diffuse_color[3]);
    v.emissive = make_float4(d->emission[0], d->emission[1], d->emission[2], 0.f);
    for (int k = 0; k < 3; k++) m.emissiveColor[k] = d->emission[k];
    pack8(v.p[2], 16, d->transmission_factor);
    pack8(v.p[2], 0, d->clearcoat_factor);
    pack8(v.p[2], 8, 1.f - d->clearcoat_roughness_factor);       // gloss = 1 - roughness (PTMaterial.cpp:176-181)
    v.transmittance.w = d->index_of_refraction;
    pack8(v.p[0], 16, d->specular_factor);
    pack8(v.p[1], 0, d->specular_tint_factor);
    pack8(v.p[0], 8, d->subsurface_factor);
    v.tint.w = d->luminance;
    pack8(v.p[1], 8, d->anisotropic);
    pack8(v.p[1], 16, d->sheen_factor);
    pack8(v.p[1], 24, d->sheen_tint_factor);
    v.tint = make_float4(d->tint_factor[0], d->tint_factor[1], d->tint_factor[2], v.tint.w);
    v.transmittance = make_float4(d->transmittance[0], d->transmittance[1], d->transmittance[2], v.transmittance.w);
    pack8(v.p[0], 24, d->roughness_factor);
    pack8(v.p[0], 0, d->metallic_factor);
    // PTMaterial::CreateDeviceMaterial (PTMaterial.cpp:97-148): the clear-coat-roughness texture overwrites the clear-coat
    // slot and the roughness slot stays a null handle; kept for parity
    v.tex[0] = tCCR; v.tex[1] = -1; v.tex[2] = tTr; v.tex[3] = tDiff; v.tex[4] = tEm; v.tex[5] = tMR; v.tex[6] = tNorm; v.tex[7] = tTint;
    for (int k = 0; k < 8; k++) {                                 // constant slots (lm_layout.h LmDevMaterial::constMask)
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (v.tex[k] >= 0) { const Texture& t = r->textures[(size_t)v.tex[k]]; if (t.w != 1u || t.h != 1u) continue; texel(t, 0, 0, c); }
        v.constMask |= 1u << k;
        v.texConst[k] = make_float4(c[0], c[1], c[2], c[3]);
    }
    // Can surface extraction produce a material outside the contracted ReSTIR evaluation?  Transmission / clear coat are factor x texel
    // (GPUExtractSurfaceData.cu:183-196): 0 when the factor byte is 0.  Roughness is texel.g x factor re-packed by truncation: the byte is
    // 0 (mirror-like: the opaque stack is absent) iff the product is below 1/255.  The prediction must be CONSERVATIVE (a surface wrongly
    // predicted "common" would be skipped by both launches): the smallest green texel goes through the SAME decode the device fetch applies
    // (sRGB table for a texture created with normalize = 1: byte 10 decodes to 0.003, not 10/255), and the bilinear filter's
    // a + t (b - a) may land an ulp below the smallest texel, hence the margin of one part in 1e3 on the threshold.
    const Texture& tmr = r->textures[(size_t)tMR];
    const float baseRough = (float)(v.p[0] >> 24) * (1.0f / 255.0f);
    const float minG = tmr.srgb ? g_srgbLut[tmr.minG] : (float)tmr.minG / 255.0f;
    m.mayBeRare = (v.p[2] & 0x00ff00ffu) != 0u || (v.p[1] & 0x0000ff00u) != 0u || minG * baseRough * 255.f < 1.001f;      // + anisotropy (a per-material constant)
    return 0;
}

int lumen_mi_create_material(lumen_mi_renderer* r, const lumen_mi_material_data* d, lumen_mi_handle* out)
{
    if (!r || !d || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    Material m;
    const int rc = fillMaterial(r, d, m);
    if (rc) return rc;
    r->materials.push_back(m);
    r->anyRareMaterial |= m.mayBeRare;
    r->materialsDirty = true;
    *out = mkh(H_MATERIAL, r->materials.size() - 1);
    return 0;
}

int lumen_mi_update_material(lumen_mi_renderer* r, lumen_mi_handle material, const lumen_mi_material_data* d)
{
    if (!r || !d) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);                                                // before any read of the resource tables (fillMaterial reads r->textures)
    size_t idx;
    if (!unh(material, H_MATERIAL, r->materials.size(), idx)) return fail(LUMEN_MI_ERR_INVALID, "bad material handle");
    Material m;
    const int rc = fillMaterial(r, d, m);
    if (rc) return rc;
    r->materials[idx] = m;
    r->anyRareMaterial |= m.mayBeRare;
    r->materialsDirty = true;
    // the emissive classification of primitives is a function of their material (FindEmissives at CreatePrimitive time in the
    // reference; re-evaluated here so that the light list follows the edit)
    for (Primitive& p : r->prims) if (p.material == idx) findEmissives(r, p);
    r->lightsDirty = true;
    return 0;
}

int lumen_mi_create_primitive(lumen_mi_renderer* r, const lumen_mi_primitive_data* d, lumen_mi_handle* out, uint32_t* numLights)
{
    if (!r || !d || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    size_t mat;
    if (!unh(d->material, H_MATERIAL, r->materials.size(), mat)) return fail(LUMEN_MI_ERR_INVALID, "bad material handle");
    if (d->n_vertices == 0 || d->n_indices < 3 || !d->index_binary || (d->index_size != 2 && d->index_size != 4)) return fail(LUMEN_MI_ERR_INVALID, "bad primitive data");
    Primitive p;
    p.material = mat;
    p.verts.resize(d->n_vertices);
    if (d->interleaved == LUMEN_MI_VERTICES_REFERENCE64) {
        // the reference's Vertex under CUDA's alignment rules (ModelStructs.h:21-28): position @0, uv @16, normal @24, tangent @48, 64 bytes
        if (!d->vertex_binary) return fail(LUMEN_MI_ERR_INVALID, "interleaved primitive without vertex_binary");
        const uint8_t* src = (const uint8_t*)d->vertex_binary;
        for (uint32_t i = 0; i < d->n_vertices; i++, src += 64) {
            memcpy(p.verts[i].pos, src, 12); memcpy(p.verts[i].uv, src + 16, 8); memcpy(p.verts[i].normal, src + 24, 12); memcpy(p.verts[i].tangent, src + 48, 16);
        }
    } else if (d->interleaved == LUMEN_MI_VERTICES_PACKED48) {
        if (!d->vertex_binary) return fail(LUMEN_MI_ERR_INVALID, "interleaved primitive without vertex_binary");
        memcpy(p.verts.data(), d->vertex_binary, (size_t)d->n_vertices * 48);
    } else if (d->interleaved != LUMEN_MI_VERTICES_SEPARATE) {
        return fail(LUMEN_MI_ERR_INVALID, "unknown vertex layout");
    } else {
        // InterleaveVertexData (WaveFrontRenderer.cpp:1091-1107): absent attributes stay zero
        if (!d->positions) return fail(LUMEN_MI_ERR_INVALID, "primitive without positions");
        memset(p.verts.data(), 0, (size_t)d->n_vertices * 48);
        for (uint32_t i = 0; i < d->n_vertices; i++) {
            memcpy(p.verts[i].pos, d->positions + 3 * i, 12);
            if (d->tex_coords) memcpy(p.verts[i].uv, d->tex_coords + 2 * i, 8);
            if (d->normals) memcpy(p.verts[i].normal, d->normals + 3 * i, 12);
            if (d->tangents) memcpy(p.verts[i].tangent, d->tangents + 4 * i, 16);
        }
    }
    p.idx.resize(d->n_indices);                                   // 16-bit indices are widened (WaveFrontRenderer.cpp:1161-1181)
    if (d->index_size == 2) { const uint16_t* s = (const uint16_t*)d->index_binary; for (uint32_t i = 0; i < d->n_indices; i++) p.idx[i] = s[i]; }
    else memcpy(p.idx.data(), d->index_binary, (size_t)d->n_indices * 4);
    for (uint32_t i : p.idx) if (i >= d->n_vertices) return fail(LUMEN_MI_ERR_INVALID, "index out of range");
    findEmissives(r, p);
    if (numLights) *numLights = p.numLights;
    r->prims.push_back(std::move(p));
    r->sceneDirty = true;
    *out = mkh(H_PRIMITIVE, r->prims.size() - 1);
    return 0;
}

int lumen_mi_create_mesh(lumen_mi_renderer* r, const lumen_mi_handle* prims, uint32_t n, lumen_mi_handle* out)
{
    if (!r || !prims || !out || n == 0) return fail(LUMEN_MI_ERR_INVALID, "bad mesh arguments");
    ApiLock lk(r);
    Mesh m;
    for (uint32_t i = 0; i < n; i++) { size_t p; if (!unh(prims[i], H_PRIMITIVE, r->prims.size(), p)) return fail(LUMEN_MI_ERR_INVALID, "bad primitive handle"); m.prims.push_back(p); }
    r->meshes.push_back(m);
    *out = mkh(H_MESH, r->meshes.size() - 1);
    return 0;
}

int lumen_mi_create_scene(lumen_mi_renderer* r, lumen_mi_handle* out)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    r->scenes.emplace_back();
    *out = mkh(H_SCENE, r->scenes.size() - 1);
    return 0;
}

int lumen_mi_set_scene(lumen_mi_renderer* r, lumen_mi_handle scene)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    size_t s;
    if (!unh(scene, H_SCENE, r->scenes.size(), s)) return fail(LUMEN_MI_ERR_INVALID, "bad scene handle");
    // the adapter calls this before every TraceFrame (include/lumen_mi_renderer.hpp): only a CHANGE of the active scene is a scene edit
    if (r->activeScene != (long)s) { r->activeScene = (long)s; r->sceneDirty = true; }
    return 0;
}

// instance handles: type tag | 24-bit generation | slot + 1 (MeshInstance objects die with ILumenScene::Clear in the reference; here
// their slots are recycled and a handle from before the clear no longer resolves)
static lumen_mi_handle instanceHandle(size_t idx, uint32_t gen) { return ((uint64_t)H_INSTANCE << 56) | ((uint64_t)(gen & 0xffffffu) << 32) | (uint64_t)(idx + 1); }
static bool instanceOf(const lumen_mi_renderer* r, lumen_mi_handle h, size_t& idx)
{
    if ((h >> 56) != (uint64_t)H_INSTANCE) return false;
    const uint64_t slot = h & 0xffffffffull;
    if (slot == 0 || slot > r->instances.size()) return false;
    idx = (size_t)slot - 1;
    const Instance& i = r->instances[idx];
    return i.alive && (uint64_t)(i.gen & 0xffffffu) == ((h >> 32) & 0xffffffull);
}

int lumen_mi_scene_add_mesh(lumen_mi_renderer* r, lumen_mi_handle scene, lumen_mi_handle mesh, lumen_mi_handle* inst)
{
    if (!r || !inst) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    size_t s, m;
    if (!unh(scene, H_SCENE, r->scenes.size(), s) || !unh(mesh, H_MESH, r->meshes.size(), m)) return fail(LUMEN_MI_ERR_INVALID, "bad scene/mesh handle");
    Instance i;
    i.scene = s; i.mesh = m;
    const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(i.M, id, sizeof id);
    i.mode = LUMEN_MI_EMISSION_ENABLED; i.radiance[0] = i.radiance[1] = i.radiance[2] = 0.f; i.scale = 1.f; i.overrideMaterial = -1;   // MeshInstance.h:24-35
    size_t slot;
    if (!r->freeInstances.empty()) { slot = r->freeInstances.back(); r->freeInstances.pop_back(); i.gen = r->instances[slot].gen + 1; r->instances[slot] = i; }
    else { slot = r->instances.size(); r->instances.push_back(i); }
    r->scenes[s].instances.push_back(slot);
    r->sceneDirty = true;
    *inst = instanceHandle(slot, r->instances[slot].gen);
    return 0;
}

int lumen_mi_scene_clear(lumen_mi_renderer* r, lumen_mi_handle scene)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    size_t s;
    if (!unh(scene, H_SCENE, r->scenes.size(), s)) return fail(LUMEN_MI_ERR_INVALID, "bad scene handle");
    for (size_t i : r->scenes[s].instances) { r->instances[i].alive = false; r->instances[i].entries.clear(); r->freeInstances.push_back(i); }
    r->scenes[s].instances.clear(); r->sceneDirty = true;
    return 0;
}

int lumen_mi_instance_set_transform(lumen_mi_renderer* r, lumen_mi_handle inst, const float m[16])
{
    if (!r || !m) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);                                                // instanceOf indexes r->instances, which add_mesh may reallocate
    size_t i;
    if (!instanceOf(r, inst, i)) return fail(LUMEN_MI_ERR_INVALID, "bad instance handle (released by lumen_mi_scene_clear?)");
    if (memcmp(r->instances[i].M, m, 64) != 0) { memcpy(r->instances[i].M, m, 64); r->transformsDirty = true; }   // polled every frame by the adapter
    return 0;
}

int lumen_mi_instance_set_emissiveness(lumen_mi_renderer* r, lumen_mi_handle inst, int mode, const float rad[3], float scale)
{
    if (!r || !rad || mode < 0 || mode > 2) return fail(LUMEN_MI_ERR_INVALID, "bad emissiveness arguments");
    ApiLock lk(r);
    size_t i;
    if (!instanceOf(r, inst, i)) return fail(LUMEN_MI_ERR_INVALID, "bad instance handle (released by lumen_mi_scene_clear?)");
    Instance& x = r->instances[i];
    if (x.mode != mode || x.radiance[0] != rad[0] || x.radiance[1] != rad[1] || x.radiance[2] != rad[2] || x.scale != scale) r->entriesDirty = true;
    x.mode = mode; x.radiance[0] = rad[0]; x.radiance[1] = rad[1]; x.radiance[2] = rad[2]; x.scale = scale;
    return 0;
}

int lumen_mi_instance_set_override_material(lumen_mi_renderer* r, lumen_mi_handle inst, lumen_mi_handle mat)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    size_t i, m = 0;
    // handle 0 clears the override: the instance falls back to the mesh's own materials, as the reference does whenever
    // m_OverrideMaterial == nullptr (PTMeshInstance.cpp:163-165)
    if (!instanceOf(r, inst, i) || (mat != 0 && !unh(mat, H_MATERIAL, r->materials.size(), m))) return fail(LUMEN_MI_ERR_INVALID, "bad handle");
    const long want = mat == 0 ? -1L : (long)m;
    if (r->instances[i].overrideMaterial != want) r->entriesDirty = true;
    r->instances[i].overrideMaterial = want;
    return 0;
}

int lumen_mi_camera_set(lumen_mi_renderer* r, const float p[3], const float right[3], const float up[3], const float fwd[3], float fov)
{
    if (!r || !p || !right || !up || !fwd) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    for (int k = 0; k < 3; k++) { r->camPos[k] = p[k]; r->camRight[k] = right[k]; r->camUp[k] = up[k]; r->camFwd[k] = fwd[k]; }
    r->fovY = fov;
    return 0;
}

int lumen_mi_set_render_resolution(lumen_mi_renderer* r, uint32_t w, uint32_t h)
{
    if (!r || w == 0 || h == 0 || w > 65535 || h > 65535) return fail(LUMEN_MI_ERR_INVALID, "bad resolution");
    std::lock_guard<std::mutex> lk(r->settingsMutex);
    r->pending.render_width = w; r->pending.render_height = h;
    r->pending.output_width = w; r->pending.output_height = h;     // WaveFrontRenderer.cpp:352
    return 0;
}
int lumen_mi_set_output_resolution(lumen_mi_renderer* r, uint32_t w, uint32_t h)
{
    if (!r || w == 0 || h == 0) return fail(LUMEN_MI_ERR_INVALID, "bad resolution");
    std::lock_guard<std::mutex> lk(r->settingsMutex);
    r->pending.output_width = w; r->pending.output_height = h;
    return 0;
}
int lumen_mi_get_render_resolution(lumen_mi_renderer* r, uint32_t* w, uint32_t* h) { if (!r || !w || !h) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); std::lock_guard<std::mutex> sl(r->settingsMutex); *w = r->pending.render_width; *h = r->pending.render_height; return 0; }
int lumen_mi_get_output_resolution(lumen_mi_renderer* r, uint32_t* w, uint32_t* h) { if (!r || !w || !h) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); std::lock_guard<std::mutex> sl(r->settingsMutex); *w = r->pending.output_width; *h = r->pending.output_height; return 0; }
int lumen_mi_set_blend_mode(lumen_mi_renderer* r, int b)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                                // blendCounter belongs to the frame state (the render thread advances it)
    { std::lock_guard<std::mutex> sl(r->settingsMutex); r->pending.blend_output = b ? 1 : 0; }
    if (b) r->blendCounter = 0;                                   // WaveFrontRenderer.cpp:377-381
    return 0;
}
int lumen_mi_get_blend_mode(lumen_mi_renderer* r, int* b) { if (!r || !b) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); std::lock_guard<std::mutex> sl(r->settingsMutex); *b = r->pending.blend_output; return 0; }
int lumen_mi_set_depth(lumen_mi_renderer* r, uint32_t d) { if (!r || d == 0 || d > LM_MAX_DEPTH) return fail(LUMEN_MI_ERR_INVALID, "depth must be in [1, 16]"); std::lock_guard<std::mutex> sl(r->settingsMutex); r->pending.depth = d; return 0; }

int lumen_mi_trace_frame_async(lumen_mi_renderer* r) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); ApiLock lk(r); return traceFrameAsync(r); }
int lumen_mi_synchronize(lumen_mi_renderer* r) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); ApiLock lk(r); return syncAndCollect(r); }
int lumen_mi_trace_frame(lumen_mi_renderer* r)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    const int rc = traceFrameAsync(r);
    if (rc) return rc;
    return syncAndCollect(r);
}
int lumen_mi_start_rendering(lumen_mi_renderer* r)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (r->renderThread.joinable()) return 0;
    r->stopFlag = false;
    r->renderThread = std::thread([r] {
        while (!r->stopFlag.load()) {
            while (r->waiters.load() > 0 && !r->stopFlag.load()) std::this_thread::yield();      // main-thread calls go first
            if (lumen_mi_trace_frame(r) != 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    });
    return 0;
}
int lumen_mi_stop_rendering(lumen_mi_renderer* r)
{
    if (!r) return 0;
    if (r->renderThread.joinable()) { r->stopFlag = true; r->renderThread.join(); }
    return 0;
}
int lumen_mi_perform_deferred_operations(lumen_mi_renderer*) { return 0; }

// read-back of a per-pixel plane of the last traced frame.  Device pointer and size are resolved UNDER the frame lock: a resolution
// change on the render thread frees and reallocates them (ensureFrameBuffers).
enum OutPlane { OUT_SRGB8, OUT_RADIANCE, OUT_DIRECT, OUT_INDIRECT, OUT_GBUFFER };
static int copyOut(lumen_mi_renderer* r, OutPlane which, void* host, size_t capacity, uint32_t* w, uint32_t* h)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    const LmFrame& f = r->fr;
    if (w) *w = f.ww; if (h) *h = f.wh;
    const void* dev = which == OUT_SRGB8 ? (const void*)f.output : which == OUT_RADIANCE ? (const void*)f.combined : which == OUT_DIRECT ? (const void*)f.direct
                    : which == OUT_INDIRECT ? (const void*)f.indirect : (const void*)f.gbuf[r->lastGbuf];
    const size_t bytes = (size_t)f.n * (which == OUT_SRGB8 ? 4u : which == OUT_GBUFFER ? 128u : 16u);
    if (!host) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (!dev || !r->allocN) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    if (capacity < bytes) return fail(LUMEN_MI_ERR_INVALID, "buffer too small");
    int rc = syncAndCollect(r);
    if (rc) return rc;
    LM_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int lumen_mi_get_output_pixels(lumen_mi_renderer* r, uint8_t* rgba8, size_t cap, uint32_t* w, uint32_t* h) { return copyOut(r, OUT_SRGB8, rgba8, cap, w, h); }
int lumen_mi_get_radiance(lumen_mi_renderer* r, float* out, size_t cap) { return copyOut(r, OUT_RADIANCE, out, cap, nullptr, nullptr); }
int lumen_mi_get_radiance_half4(lumen_mi_renderer* r, uint16_t* out, size_t cap)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    const uint32_t n = r->fr.n;
    if (!r->fr.combined || !r->allocN) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    if (cap < (size_t)n * 8) return fail(LUMEN_MI_ERR_INVALID, "buffer too small");
    int rc = syncAndCollect(r); if (rc) return rc;
    DevBuf<uint2>& d = r->dExportHalf;               // kept with the renderer (released with the frame buffers): no allocation per call, nothing to leak on an error return
    if (d.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "export allocation failed");
    r->K->export_half4(r->stream, r->gridFor(n, 8), r->fr.combined, d.p, n);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}
int lumen_mi_get_channel(lumen_mi_renderer* r, int ch, float* out, size_t cap)
{
    if (!r || ch < 0 || ch > 1) return fail(LUMEN_MI_ERR_INVALID, "channel must be 0 (DIRECT) or 1 (INDIRECT)");
    return copyOut(r, ch == 0 ? OUT_DIRECT : OUT_INDIRECT, out, cap, nullptr, nullptr);
}
int lumen_mi_copy_radiance_device(lumen_mi_renderer* r, void* dst)
{
    if (!r || !dst) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    if (!r->fr.combined) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    LM_HIP(hipMemcpyAsync(dst, r->fr.combined, (size_t)r->fr.n * 16, hipMemcpyDeviceToDevice, r->stream));
    return 0;
}
// multi-GPU tiles (tiles.py TileGather): the rectangle [x0, x1) x [y0, y1) of the IMAGE (inside the render window) out of the merged radiance into a pitched device
// image, and a pitched rectangle between two device images — both on the renderer's stream, both by a kernel of this library (nothing is loaded at the first call)
int lumen_mi_copy_radiance_rect_device(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* dst, uint32_t dstPitch)
{
    if (!r || !dst) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    if (!r->fr.combined || !r->fr.n) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    const LmFrame& f = r->fr;
    if (x0 >= x1 || y0 >= y1 || x0 < f.x0 || y0 < f.y0 || x1 > f.x0 + f.ww || y1 > f.y0 + f.wh) return fail(LUMEN_MI_ERR_INVALID, "rectangle outside the render window");
    if (dstPitch < x1 - x0) return fail(LUMEN_MI_ERR_INVALID, "destination pitch smaller than the rectangle");
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t w = x1 - x0, h = y1 - y0;
    r->K->copy_rect(r->stream, r->gridFor(w * h, 8), (float4*)dst, dstPitch, f.combined + (size_t)(y0 - f.y0) * f.ww + (x0 - f.x0), f.ww, w, h);
    LM_HIP(hipGetLastError());
    return 0;
}
int lumen_mi_copy_rect_device(lumen_mi_renderer* r, void* dst, uint32_t dstPitch, const void* src, uint32_t srcPitch, uint32_t w, uint32_t h)
{
    if (!r || !dst || !src) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (!w || !h || dstPitch < w || srcPitch < w) return fail(LUMEN_MI_ERR_INVALID, "empty rectangle or pitch smaller than the rectangle");
    if ((uint64_t)w * h > 0xffffffffull) return fail(LUMEN_MI_ERR_INVALID, "rectangle of 2^32 pixels or more (the copy kernel indexes pixels with 32 bits)");
    ApiLock lk(r);
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    r->K->copy_rect(r->stream, r->gridFor(w * h, 8), (float4*)dst, dstPitch, (const float4*)src, srcPitch, w, h);
    LM_HIP(hipGetLastError());
    return 0;
}
// multi-GPU seams: see lm_k_history_copy (kernels.hip) and tiles.py exchange_history
static int historyCopy(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* dev, int import)
{
    if (!r || !dev) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    if (!r->fr.combined || !r->fr.n || !r->allocN) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");      // allocN == 0: a failed reallocation left no frame buffers
    const LmFrame& f = r->fr;
    if (x0 >= x1 || y0 >= y1 || x0 < f.x0 || y0 < f.y0 || x1 > f.x0 + f.ww || y1 > f.y0 + f.wh) return fail(LUMEN_MI_ERR_INVALID, "rectangle outside the render window");
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t w = x1 - x0, h = y1 - y0;
    // the last frame may have left its history passes pending (lazy reuse, frame.cpp): when the swap chain has turned, what is copied here is their result
    if (r->owed.valid) { launchOwedReuse(r, r->stream, 2); r->K->reuse_settle(r->stream, f); }
    r->K->history_copy(r->stream, r->gridFor(w * h, 8), f, x0 - f.x0, y0 - f.y0, w, h, (float4*)dev, import);
    LM_HIP(hipGetLastError());
    return 0;
}
int lumen_mi_export_history(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* device_dst) { return historyCopy(r, x0, y0, x1, y1, device_dst, 0); }
int lumen_mi_import_history(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const void* device_src) { return historyCopy(r, x0, y0, x1, y1, (void*)device_src, 1); }

static int waveSync(lumen_mi_renderer* r, void* dev, int import)
{
    if (!r || !dev) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    if (!r->fr.swap) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    r->K->wave_sync(r->stream, r->fr.swap, (int*)dev, import);
    LM_HIP(hipGetLastError());
    return 0;
}
int lumen_mi_export_wave_count(lumen_mi_renderer* r, void* device_i32) { return waveSync(r, device_i32, 0); }
int lumen_mi_import_wave_count(lumen_mi_renderer* r, const void* device_i32) { return waveSync(r, (void*)device_i32, 1); }

int lumen_mi_get_gbuffer(lumen_mi_renderer* r, float* out, size_t cap)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    return copyOut(r, OUT_GBUFFER, out, cap, nullptr, nullptr);
}

int lumen_mi_get_denoiser_inputs(lumen_mi_renderer* r, float minD, float maxD, float* depth, uint16_t* normalRoughness, uint16_t* motion)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    int rc = syncAndCollect(r); if (rc) return rc;
    const uint32_t n = r->fr.n;
    if (!n || !r->fr.gbuf[0]) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced");
    const int last = r->lastGbuf;
    DevBuf<float> dDepth; DevBuf<uint2> dNr;
    if ((depth && dDepth.ensure(n)) || (normalRoughness && dNr.ensure(n))) return fail(LUMEN_MI_ERR_DEVICE, "export allocation failed");
    if (normalRoughness) LM_HIP(hipMemsetAsync(dNr.p, 0, (size_t)n * sizeof(uint2), r->stream));
    r->K->export_aux(r->stream, r->gridFor(n, 8), r->fr, last, minD, maxD, depth ? dDepth.p : nullptr, normalRoughness ? dNr.p : nullptr);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    if (depth) LM_HIP(hipMemcpy(depth, dDepth.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (normalRoughness) LM_HIP(hipMemcpy(normalRoughness, dNr.p, (size_t)n * sizeof(uint2), hipMemcpyDeviceToHost));
    if (motion) LM_HIP(hipMemcpy(motion, r->fr.motion, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    dDepth.release(); dNr.release();
    return 0;
}

int lumen_mi_get_frame_stat(lumen_mi_renderer* r, const char* key, uint64_t* us)
{
    if (!r || !key || !us) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (std::string(key) == "Frames Traced") { *us = r->framesTraced.load(); return 0; }      // TraceFrames enqueued since the renderer was created (FrameStats::m_Id of the adapter)
    ApiLock lk(r);                                  // the render thread rewrites the map when it collects a frame's events
    auto it = r->frameStats.find(key);
    if (it == r->frameStats.end()) return fail(LUMEN_MI_ERR_INVALID, std::string("no such frame-stat key: ") + key);
    *us = it->second;
    return 0;
}
int lumen_mi_get_counters(lumen_mi_renderer* r, uint64_t* out, uint32_t n)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    { ApiLock lk(r); int rc = syncAndCollect(r); if (rc) return rc; }
    uint64_t v[64] = {0};
    const uint32_t* c = r->hostCounters;
    for (uint32_t d = 0; d < r->lastDepth && d < 16; d++) { v[0] += c[LM_CNT_RAYS(d)]; v[4 + d] = c[LM_CNT_RAYS(d)]; v[1] += c[LM_CNT_SHADOW(d)]; }
    v[2] = (uint64_t)c[LM_CNT_RESTIR(0)] + c[LM_CNT_RESTIR(1)];
    v[3] = r->lastLightCount;
    v[22] = (uint64_t)c[LM_CNT_NODES] | ((uint64_t)c[LM_CNT_NODES + 1] << 32);     // child boxes slab-tested
    v[20] = v[22] / 2;                                                                 // = binary-node equivalents (2 boxes per node)
    v[21] = (uint64_t)c[LM_CNT_TRIS] | ((uint64_t)c[LM_CNT_TRIS + 1] << 32);
    for (int k = 0; k < 16; k++) v[24 + k] = c[LM_CNT_STEP_HIST + k];
    v[40] = c[LM_CNT_STEP_MAX];
    if (r->instrumented) {     // stack pushes of the counting build: total in LDS / in the global spill area since the last call
        DevBuf<unsigned long long> d; unsigned long long h[2] = {0, 0};
        if (!d.ensure(2)) { lm_read_pushes(r->stream, d.p); if (hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) { v[45] = h[0]; v[46] = h[1]; } d.release(); }
    }
    for (int k = 0; k < 4; k++) v[41 + k] = (uint64_t)c[LM_CNT_OCC + 2 * k] | ((uint64_t)c[LM_CNT_OCC + 2 * k + 1] << 32);
    v[48] = c[LM_CNT_RESTIR(0)]; v[49] = c[LM_CNT_RESTIR(1)];
    v[50] = r->refits; v[51] = r->assemblies;                                          // GPU refits / instance-level assemblies since creation
    if (r->fr.swap) { int dv[2] = {0, 0}; if (hipMemcpy(dv, r->fr.swap + 8, sizeof dv, hipMemcpyDeviceToHost) == hipSuccess) { v[54] = (uint64_t)dv[0]; v[55] = (uint64_t)dv[1]; } }     // lazy reuse: deferred history passes that ran / entries completed instead
    v[56] = r->gpuBuilds;
    v[52] = c[LM_CNT_RARE]; v[53] = r->anyRareMaterial ? 1u : 0u;                      // depth-0 surfaces outside the contracted ReSTIR evaluation / can any material produce one
    for (uint32_t i = 0; i < n && i < 64; i++) out[i] = v[i];
    return 0;
}
int lumen_mi_get_counter_totals(lumen_mi_renderer* r, uint64_t* out, uint32_t n, int reset)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = syncAndCollect(r); if (rc) return rc;
    uint64_t v[64] = {0};
    if (r->dTotals.p) {
        unsigned long long t[LM_CNT_WORDS + 1];
        LM_HIP(hipMemcpy(t, r->dTotals.p, sizeof t, hipMemcpyDeviceToHost));
        // every depth slot, not only those of the last frame's depth: SetDepth may have lowered the depth inside the window that is summed (unused slots are zero)
        for (uint32_t d = 0; d < LM_MAX_DEPTH; d++) { v[0] += t[LM_CNT_RAYS(d)]; v[4 + d] = t[LM_CNT_RAYS(d)]; v[1] += t[LM_CNT_SHADOW(d)]; }
        v[2] = t[LM_CNT_RESTIR(0)] + t[LM_CNT_RESTIR(1)];
        v[3] = t[LM_CNT_WORDS];
        v[48] = t[LM_CNT_RESTIR(0)]; v[49] = t[LM_CNT_RESTIR(1)];
        if (reset) LM_HIP(hipMemsetAsync(r->dTotals.p, 0, sizeof t, r->stream));
    }
    for (uint32_t i = 0; i < n && i < 64; i++) out[i] = v[i];
    return 0;
}
int lumen_mi_get_kernel_time(lumen_mi_renderer* r, int which, float* ms, uint32_t* launches)
{
    if (!r || which < 0 || which > 5) return fail(LUMEN_MI_ERR_INVALID, "bad kernel class");
    if (ms) *ms = r->classMs[which]; if (launches) *launches = r->classLaunches[which];
    return 0;
}
int lumen_mi_enable_kernel_timing(lumen_mi_renderer* r, int e)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    if (e) { for (int c = 0; c < 6; c++) { r->classMs[c] = 0.f; r->classLaunches[c] = 0; } }      // enabling starts a new accumulation window
    r->timing = e < 0 ? 0 : (e > 2 ? 1 : e);
    return 0;
}
int lumen_mi_set_tuning(lumen_mi_renderer* r, const char* key, int value)
{
    if (!r || !key) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    const std::string k = key;
    if (k == "tail_below") r->tailBelow = value;
    else if (k == "wave_streams") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->waveStreams = value == 2 ? 2 : 1; }
    else if (k == "tail_lanes") r->tailLanes = value <= 0 ? -1 : std::min(64, value);
    else if (k == "tail_pair") r->tailPair = value;
    else if (k == "single_stream") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->overlap = value == 0; }
    else if (k == "refit") r->refitEnabled = value;
    else if (k == "pick_ahead") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->pickAhead = value; }
    else if (k == "fuzz") r->fuzz = (uint32_t)value;
    else if (k == "assemble") r->assembleEnabled = value;
    else if (k == "shadow_on_wave") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->shadowOnWave = value; }
    else if (k == "fast_resample") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->fastResample = value != 0; applyNoSlpKernels(r); }
    else if (k == "packet_primary") r->packetPrimary = value;
    else if (k == "packet_visibility") r->packetVisibility = value;
    else if (k == "spatial_lds") r->spatialLds = value;
    else if (k == "fuse_combine") r->fuseCombine = value;
    else if (k == "pick_wide") { if (value < 0 || value > 2) return fail(LUMEN_MI_ERR_INVALID, "pick_wide: 0, 1 or 2"); r->pickWide = value; }
    else if (k == "trace_blocks_main" || k == "trace_blocks_vis" || k == "trace_blocks_aux") {
        if (value < 0 || value > 8 || (value == 0 && k == "trace_blocks_aux")) return fail(LUMEN_MI_ERR_INVALID, "trace_blocks_*: 1 .. 8 blocks per CU (main / vis: 0 = chosen per frame)");
        (k == "trace_blocks_main" ? r->traceBlocksMain : k == "trace_blocks_vis" ? r->traceBlocksVis : r->traceBlocksAux) = value;
    }
    else if (k == "tail_repack") r->tailRepack = value;
    else if (k == "gpu_build") { if (r->gpuBuild != value) { r->gpuBuild = value; r->sceneDirty = true; r->builtOnce = false; } }      // takes effect with a full rebuild at the next frame
    else if (k == "lazy_reuse") r->lazyReuse = value;
    else if (k == "fuse_primary") r->fusePrimary = value;
    else if (k == "fast_shade") r->fastShade = value;
    else if (k == "sort_rays") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->sortRays = std::max(0, value); }
    else if (k == "tex_filter") {                         // every emissive-triangle verdict and the texture descriptors depend on it
        if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; }
        if (r->texFilter != (value != 0)) { r->texFilter = value != 0; r->texturesDirty = true; for (Primitive& p : r->prims) findEmissives(r, p); r->lightsDirty = true; r->sceneDirty = true; }
    }
    else if (k == "refill") r->refillBelow = value;
    else if (k == "refill_visibility") r->refillVisibility = value;
    else if (k == "refill_primary") r->refillPrimary = value;
    else return fail(LUMEN_MI_ERR_INVALID, std::string("unknown tuning key: ") + key);
    return 0;
}
int lumen_mi_set_instrumented(lumen_mi_renderer* r, int e) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); ApiLock lk(r); r->instrumented = e != 0; r->K = e ? lm_kernel_table_instrumented() : &r->Kmix; return 0; }

int lumen_mi_set_tile(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    if (x0 >= x1 || y0 >= y1) { r->tileSet = false; return 0; }          // an empty rectangle: the whole window is owned again
    r->ox0 = x0; r->oy0 = y0; r->ox1 = x1; r->oy1 = y1; r->tileSet = true;
    return 0;
}

int lumen_mi_set_window(lumen_mi_renderer* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { ApiLock lk(r); r->windowSet = false; return 0; }      // back to the whole image
    if (x0 >= x1 || y0 >= y1) return fail(LUMEN_MI_ERR_INVALID, "empty window");
    ApiLock lk(r);
    r->wx0 = x0; r->wy0 = y0; r->wx1 = x1; r->wy1 = y1; r->windowSet = true;
    return 0;
}

static int prepareScene(lumen_mi_renderer* r)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    int rc;
    if ((rc = uploadResources(r))) return rc;
    if ((rc = flatten(r))) return rc;
    // outside a frame: every frame in flight has to be complete before a scene set is rewritten (the merge on the main stream joins
    // the other streams)
    LM_HIP(hipStreamSynchronize(r->stream));
    if ((rc = syncScene(r, r->stream))) return rc;
    LM_HIP(hipStreamSynchronize(r->stream));                    // the refit scratch is shared with the refits of later frames
    if ((rc = r->dCounters.ensure(2 * LM_CNT_WORDS))) return fail(LUMEN_MI_ERR_DEVICE, "counter allocation failed");
    return 0;
}

int lumen_mi_query_closest(lumen_mi_renderer* r, uint32_t n, const float* o, const float* d, float tmin, float tmax, uint32_t* ip, float* uvt)
{
    if (!r || !o || !d || !ip || !uvt) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    std::vector<float4> ho(n), hd(n);
    for (uint32_t i = 0; i < n; i++) { ho[i] = make_float4(o[3*i], o[3*i+1], o[3*i+2], 0.f); hd[i] = make_float4(d[3*i], d[3*i+1], d[3*i+2], 0.f); }
    DevBuf<float4> dO, dD, dU; DevBuf<uint4> dI;
    if (dO.upload(ho, r->stream) || dD.upload(hd, r->stream) || dU.ensure(n) || dI.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "query allocation failed");
    // counting build: the traversal statistics of THIS query (steps per ray, longest ray, lane occupancy) replace the last frame's in lumen_mi_get_counters
    if (r->instrumented) LM_HIP(hipMemsetAsync(r->dCounters.p, 0, LM_CNT_WORDS * sizeof(uint32_t), r->stream));
    r->K->query_closest(r->stream, r->traceGrid(), r->dscene, dO.p, dD.p, n, tmin, tmax, dI.p, dU.p, r->dCounters.p);
    std::vector<uint4> hi(n); std::vector<float4> hu(n);
    LM_HIP(hipStreamSynchronize(r->stream));
    if (r->instrumented) { LM_HIP(hipMemcpy(r->hostCounters, r->dCounters.p, sizeof r->hostCounters, hipMemcpyDeviceToHost)); r->countersValid = true; }
    LM_HIP(hipMemcpy(hi.data(), dI.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    LM_HIP(hipMemcpy(hu.data(), dU.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { ip[2*i] = hi[i].x; ip[2*i+1] = hi[i].y; uvt[3*i] = hu[i].x; uvt[3*i+1] = hu[i].y; uvt[3*i+2] = hu[i].z; }
    dO.release(); dD.release(); dU.release(); dI.release();
    return 0;
}
int lumen_mi_query_any(lumen_mi_renderer* r, uint32_t n, const float* o, const float* d, float tmin, const float* tmax, uint8_t* occ)
{
    if (!r || !o || !d || !tmax || !occ) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    std::vector<float4> ho(n), hd(n);
    for (uint32_t i = 0; i < n; i++) { ho[i] = make_float4(o[3*i], o[3*i+1], o[3*i+2], tmax[i]); hd[i] = make_float4(d[3*i], d[3*i+1], d[3*i+2], 0.f); }
    DevBuf<float4> dO, dD; DevBuf<uint32_t> dR;
    if (dO.upload(ho, r->stream) || dD.upload(hd, r->stream) || dR.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "query allocation failed");
    r->K->query_any(r->stream, r->traceGrid(), r->dscene, dO.p, dD.p, n, tmin, dR.p, r->dCounters.p);
    std::vector<uint32_t> hr(n);
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(hr.data(), dR.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) occ[i] = (uint8_t)hr[i];
    dO.release(); dD.release(); dR.release();
    return 0;
}

#ifndef LUMEN_MI_TEST_HOOKS
#define LUMEN_MI_TEST_HOOKS 1
#endif
#if LUMEN_MI_TEST_HOOKS      // known-answer hooks (test surface; `make HOOKS=0` leaves them out: csrc/lm_hooks.h)
int lumen_mi_test_bsdf(lumen_mi_renderer* r, uint32_t n, int mode, const float* mat23, const float* N, const float* T, const float* wo, const float* aux, float* out8)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!mat23 || !N || !T || !wo || !aux || !out8) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);                                  // r->stream is the render thread's stream too
    LM_HIP(hipSetDevice(r->device));
    DevBuf<float> dm, dn, dt, dw, da, dout;
    std::vector<float> vm(mat23, mat23 + (size_t)23 * n), vn(N, N + (size_t)3 * n), vt(T, T + (size_t)3 * n), vw(wo, wo + (size_t)3 * n), va(aux, aux + (size_t)3 * n);
    if (dm.upload(vm, r->stream) || dn.upload(vn, r->stream) || dt.upload(vt, r->stream) || dw.upload(vw, r->stream) || da.upload(va, r->stream) || dout.ensure((size_t)8 * n)) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    r->K->test_bsdf(r->stream, n, mode, dm.p, dn.p, dt.p, dw.p, da.p, dout.p);
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out8, dout.p, (size_t)8 * n * 4, hipMemcpyDeviceToHost));
    dm.release(); dn.release(); dt.release(); dw.release(); da.release(); dout.release();
    return 0;
}
int lumen_mi_test_camera(const float right[3], const float up[3], const float forward[3], const float prev_world16[16], float fov, float aspect, float out25[25])
{
    if (!right || !up || !forward || !prev_world16 || !out25) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    cameraVectors(right, up, forward, fov, aspect, out25, out25 + 3, out25 + 6);
    motionMatrix(prev_world16, fov, aspect, out25 + 9);
    return 0;
}
int lumen_mi_test_restir(lumen_mi_renderer* r, int mode, uint32_t n, const float* a, const float* b, const uint32_t* c, uint32_t m, float* out)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    const bool resample = mode == 3 || mode == 5, combine = mode == 4 || mode == 6;
    if (mode < 0 || mode > 6 || !a || !out || n == 0 || (mode == 0 && (!b || !c)) || (mode == 1 && (!b || m == 0)) || (resample && !b) || (combine && (!b || !c)))
        return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    ApiLock lk(r);                                  // r->stream is the render thread's stream too
    LM_HIP(hipSetDevice(r->device));
    const size_t na = mode == 0 ? (size_t)8 * n : (resample || combine) ? (size_t)35 * n : n;
    const size_t nb = mode == 0 ? (size_t)8 * n : mode == 1 ? m : resample ? (size_t)14 * n : combine ? (size_t)34 * n : 0;
    const size_t nc = mode == 0 ? (size_t)8 * n : combine ? n : 0;
    const size_t nout = mode == 0 ? (size_t)33 * n : mode == 1 ? (size_t)2 * m : resample ? (size_t)5 * n : combine ? (size_t)18 * n : n;
    DevBuf<float> da, db, dout; DevBuf<uint32_t> dc;
    struct Release { DevBuf<float>&a, &b, &o; DevBuf<uint32_t>& c; ~Release() { a.release(); b.release(); o.release(); c.release(); } } guard{da, db, dout, dc};   // on every exit path
    std::vector<float> va(a, a + na), vb; std::vector<uint32_t> vc;
    if (nb) vb.assign(b, b + nb);
    if (nc) vc.assign(c, c + nc);
    if (da.upload(va, r->stream) || db.upload(vb, r->stream) || dc.upload(vc, r->stream) || dout.ensure(nout)) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    r->K->test_restir(r->stream, mode, n, da.p, db.p, dc.p, m, dout.p);
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out, dout.p, nout * 4, hipMemcpyDeviceToHost));
    return 0;
}
int lumen_mi_test_math(lumen_mi_renderer* r, uint32_t n, int fn, const float* x, const float* y, float* out)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!x || !y || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    LM_HIP(hipSetDevice(r->device));
    DevBuf<float> dx, dy, dout;
    std::vector<float> vx(x, x + n), vy(y, y + n);
    if (dx.upload(vx, r->stream) || dy.upload(vy, r->stream) || dout.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    r->K->test_math(r->stream, n, fn, dx.p, dy.p, dout.p);
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    dx.release(); dy.release(); dout.release();
    return 0;
}

// host-only scene products (no device needed beyond what flatten uploads)
#endif   // LUMEN_MI_TEST_HOOKS

int lumen_mi_get_world_triangles(lumen_mi_renderer* r, float* out, uint32_t cap, uint32_t* count)
{
    if (!r || !count) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    *count = (uint32_t)r->triEntry.size();
    if (out) { if (cap < *count) return fail(LUMEN_MI_ERR_INVALID, "buffer too small"); ensureWorldTris(r); memcpy(out, r->worldTris.data(), r->worldTris.size() * 4); }
    return 0;
}
int lumen_mi_get_lights(lumen_mi_renderer* r, float* lights16, float* cdf, uint32_t cap, uint32_t* count)
{
    if (!r || !count) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    if ((rc = buildLights(r))) return rc;
    *count = (uint32_t)r->lights.size();
    if (lights16 || cdf) {
        if (cap < *count) return fail(LUMEN_MI_ERR_INVALID, "buffer too small");
        if (lights16) memcpy(lights16, r->lights.data(), r->lights.size() * sizeof(LmLight));
        if (cdf) memcpy(cdf, r->cdf.data(), r->cdf.size() * 4);
    }
    return 0;
}
int lumen_mi_get_bvh_info(lumen_mi_renderer* r, uint32_t* nodes, uint32_t* tris, uint32_t* maxDepth)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    // binary nodes of the SAH build; 4-wide nodes of an assembled tree
    if (nodes) *nodes = (uint32_t)(r->bvh.nodes.empty() ? r->bvh.nodesW.size() : r->bvh.nodes.size());
    if (tris) *tris = (uint32_t)r->bvh.order.size();
    if (maxDepth) *maxDepth = r->bvh.maxDepth;
    return 0;
}

}  // extern "C"
