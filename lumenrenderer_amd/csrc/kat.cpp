// kat.cpp — known-answer hooks that run whole KERNELS of the hot path on rows (C ABI: lumen_mi_test_restir_frame, lumen_mi_test_shade, lumen_mi_test_primary_rays).
//
// The rows (tests/golden/ref_kat5.npz) are what the reference's own __global__ kernel bodies computed on a synthetic image (ReSTIRKernels.cu:343-370, 402-522,
// 546-582, 600-616, 787-980, 1015-1121, 1407-1436; GPUShadeDirect.cu:42-153; GPUShadeIndirect.cu:7-146; GPUGeneratePrimRay.cu:28-82).  Nothing here computes:
// synthetic surfaces and reservoirs are laid out by the product's own store functions (kernels.hip lm_k_kat_*), the product's kernels are launched through the
// kernel table with the arguments, order and seed evolution of frame.cpp (Framework/ReSTIR.cpp:65-233), and the buffers are read back through the product's load
// functions.  The only stand-in is the tracer: the reference's visibility programs are closed (OptiX), so the rows carry an occlusion mask per pass and
// lm_k_kat_resolve hands it to the same lines the traversal kernels run on a resolved ray (lm_vis_resolve).
#ifndef LUMEN_MI_TEST_HOOKS
#define LUMEN_MI_TEST_HOOKS 1
#endif
#if LUMEN_MI_TEST_HOOKS      // the whole file is test surface: `make HOOKS=0` builds the library without it (csrc/lm_hooks.h)
#include "renderer_state.h"

using namespace lmr;

namespace {
struct Bufs {                         // everything freed on every exit path
    std::vector<void*> all;
    template <class T> T* get(size_t count, bool zero = true)
    {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
        all.push_back(p);
        if (zero && hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
        return (T*)p;
    }
    template <class T> T* put(const T* host, size_t count)
    {
        T* p = get<T>(count, false);
        if (p && count && hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return p;
    }
    ~Bufs() { for (void* p : all) (void)hipFree(p); }
};
LmLight* putLights(Bufs& b, const uint32_t* lights16, uint32_t n) { static_assert(sizeof(LmLight) == 64, "16 words per light"); return (LmLight*)b.put<uint32_t>(lights16, (size_t)16 * n); }
float wordToFloat(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
}  // namespace

extern "C" {

int lumen_mi_test_restir_frame(lumen_mi_renderer* r, uint32_t W, uint32_t H, const uint32_t* surf_cur40, const uint32_t* surf_prev40, const uint32_t* motion_half2,
                               uint32_t n_lights, const uint32_t* lights16, const uint32_t* cdf, uint32_t a_seed, int current_index, const uint8_t* occluded0,
                               const uint8_t* occluded1, int fast, uint32_t* res4, uint32_t* bags, uint32_t* stages, uint32_t* rays, uint32_t* ray_counts, uint32_t* direct)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                        // before any renderer state is read (a render thread may be running)
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!W || !H || !surf_cur40 || !motion_half2 || !n_lights || !lights16 || !cdf || !occluded0 || !occluded1 || !res4 || !stages || !rays || !ray_counts || !direct ||
        (current_index != 0 && current_index != 1) || fast < 0 || fast > 2) return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    LM_HIP(hipSetDevice(r->device));
    LM_HIP(hipStreamSynchronize(r->stream));
    const uint32_t n = W * H;
    const LmKernelTable* K = r->K;                        // the table lumen_mi_set_instrumented selected
    hipStream_t st = r->stream;
    Bufs b;
    LmFrame fr{};
    fr.W = W; fr.H = H; fr.x0 = 0; fr.y0 = 0; fr.ww = W; fr.wh = H; fr.tx0 = 0; fr.ty0 = 0; fr.tx1 = W; fr.ty1 = H; fr.n = n;
    bool ok = true;
    for (int i = 0; i < 2; i++) ok &= (fr.gbuf[i] = b.get<float4>((size_t)8 * n)) && (fr.probe[i] = b.get<float4>(n));      // zero-filled: the previous buffer of a first frame
    for (int i = 0; i < 4; i++) ok &= (fr.res[i] = b.get<float4>((size_t)4 * n)) && (fr.resC[i] = b.get<float4>(n));
    ok &= (fr.visO = b.get<float4>(n)) && (fr.visD = b.get<float4>(n)) && (fr.vis2O = b.get<float4>(n)) && (fr.vis2D = b.get<float4>(n));
    ok &= (fr.reuseMask = b.get<uint32_t>(n)) && (fr.direct = b.get<float4>(n)) && (fr.indirect = b.get<float4>(n)) && (fr.counters = b.get<uint32_t>(LM_CNT_WORDS));
    ok &= (fr.bags = b.get<uint2>(50u * 1000u)) != nullptr;
    std::vector<uint32_t> mv(n);
    for (uint32_t i = 0; i < n; i++) mv[i] = (motion_half2[2u * i] & 0xffffu) | (motion_half2[2u * i + 1u] << 16);
    ok &= (fr.motion = b.put<uint32_t>(mv.data(), n)) != nullptr;
    // swap chain: [0] front buffer; [2], [3] both buffers count as written (the previous one is read, whatever it holds); [5] nothing owed
    int swapHost[16] = {0}; swapHost[0] = current_index; swapHost[2] = 1; swapHost[3] = 1; swapHost[4] = current_index; swapHost[5] = 1;
    ok &= (fr.swap = b.put<int>(swapHost, 16)) != nullptr;
    fr.deferred = 0; fr.owedSet = -1; fr.hazardList = nullptr;
    LmScene sc{};
    ok &= (sc.lights = putLights(b, lights16, n_lights)) != nullptr;
    ok &= (sc.cdf = (const float*)b.put<uint32_t>(cdf, n_lights)) != nullptr;
    sc.numLights = n_lights; sc.cdfSum = wordToFloat(cdf[n_lights - 1u]);
    uint32_t* dRows = b.put<uint32_t>(surf_cur40, (size_t)40 * n);
    uint32_t* dRowsPrev = surf_prev40 ? b.put<uint32_t>(surf_prev40, (size_t)40 * n) : nullptr;
    uint32_t* dRes = b.put<uint32_t>(res4, (size_t)4 * n * 17);
    uint32_t* dStages = b.get<uint32_t>((size_t)5 * n * 17);
    uint8_t* dOcc0 = b.put<uint8_t>(occluded0, n); uint8_t* dOcc1 = b.put<uint8_t>(occluded1, n);
    ok &= dRows && (dRowsPrev || !surf_prev40) && dRes && dStages && dOcc0 && dOcc1;
    if (!ok) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    if (fast) { const uint32_t one = 1u; LM_HIP(hipMemcpy(fr.counters + LM_CNT_RARE, &one, 4, hipMemcpyHostToDevice)); }     // the second (exact) launch of the fast mode always runs here

    K->kat_pack_surfaces(st, dRows, n, fr.gbuf[0], fr.probe[0]);
    if (dRowsPrev) K->kat_pack_surfaces(st, dRowsPrev, n, fr.gbuf[1], fr.probe[1]);
    for (int i = 0; i < 4; i++) K->kat_reservoirs(st, dRes + (size_t)i * n * 17, n, fr.res[i], fr.resC[i], 0);
    auto tap = [&](int stage, int buf) { K->kat_reservoirs(st, dStages + (size_t)stage * n * 17, n, fr.res[buf], fr.resC[buf], 1); };
    const int cur = 0, prev = 1;                                                                      // G-buffer sets
    const int tiles = (int)(((W + 15u) / 16u) * ((H + 15u) / 16u));
    // ---- ReSTIR::Run as frame.cpp enqueues it (history passes with their frame: lazy reuse off)
    uint32_t rs = wangHash(a_seed);
    K->fill_bags(st, sc, fr, a_seed, 50u * 1000u);
    rs = wangHash(rs);
    K->pick_primary(st, tiles, sc, fr, cur, LM_RES_CUR, rs, fr.counters + LM_CNT_RESTIR(0), fast);   // + visibility rays, pass 1
    tap(0, current_index);
    K->kat_resolve(st, fr, LM_RES_CUR, fr.counters + LM_CNT_RESTIR(0), dOcc0, 0);
    rs = wangHash(rs);
    K->temporal(st, tiles, fr, cur, prev, LM_RES_CUR, LM_RES_PREV, LM_RES_CUR, rs, fr.counters + LM_CNT_RESTIR(1), fast);   // + visibility rays, pass 2
    tap(1, current_index);
    rs = wangHash(rs);
    K->spatial(st, tiles, fr, cur, LM_RES_CUR, 2, rs, 30, 0, fast);
    tap(2, 2);
    K->kat_resolve(st, fr, LM_RES_CUR, fr.counters + LM_CNT_RESTIR(1), dOcc1, 1);
    K->spatial(st, tiles, fr, cur, 2, 3, rs, 0, 1, fast);
    tap(3, 3);
    K->combine(st, tiles, fr, cur, LM_RES_CUR, 3, wangHash(rs), fast);
    tap(4, current_index);
    for (int i = 0; i < 4; i++) K->kat_reservoirs(st, dRes + (size_t)i * n * 17, n, fr.res[i], fr.resC[i], 1);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(st));

    LM_HIP(hipMemcpy(res4, dRes, (size_t)4 * n * 17 * 4, hipMemcpyDeviceToHost));
    LM_HIP(hipMemcpy(stages, dStages, (size_t)5 * n * 17 * 4, hipMemcpyDeviceToHost));
    LM_HIP(hipMemcpy(direct, fr.direct, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (bags) LM_HIP(hipMemcpy(bags, fr.bags, (size_t)50 * 1000 * 8, hipMemcpyDeviceToHost));
    uint32_t counts[LM_CNT_WORDS];
    LM_HIP(hipMemcpy(counts, fr.counters, sizeof counts, hipMemcpyDeviceToHost));
    std::vector<float4> qo(n), qd(n);
    for (int p = 0; p < 2; p++) {
        const uint32_t cnt = counts[LM_CNT_RESTIR(p)];
        if (cnt > n) return fail(LUMEN_MI_ERR_DEVICE, "visibility queue overflow");
        ray_counts[p] = cnt;
        LM_HIP(hipMemcpy(qo.data(), p ? fr.vis2O : fr.visO, (size_t)cnt * 16, hipMemcpyDeviceToHost));
        LM_HIP(hipMemcpy(qd.data(), p ? fr.vis2D : fr.visD, (size_t)cnt * 16, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < cnt; k++) {       // queue record -> RestirShadowRay (ReSTIRData.h:71-77): index, origin, direction, distance
            uint32_t* o = rays + ((size_t)p * n + k) * 8u;
            memcpy(o, &qd[k].w, 4);
            memcpy(o + 1, &qo[k].x, 12); memcpy(o + 4, &qd[k].x, 12); memcpy(o + 7, &qo[k].w, 4);
        }
    }
    return 0;
}

int lumen_mi_test_shade(lumen_mi_renderer* r, uint32_t n, uint32_t W, uint32_t H, const uint32_t* rows43, uint32_t n_lights, const uint32_t* lights16, const uint32_t* cdf,
                        int fast, uint32_t* direct12, uint32_t* indirect10)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                        // before any renderer state is read
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!n || !W || !H || !rows43 || !n_lights || !lights16 || !cdf || (!direct12 && !indirect10)) return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    LM_HIP(hipSetDevice(r->device));
    Bufs b;
    LmScene sc{};
    sc.lights = putLights(b, lights16, n_lights);
    sc.cdf = (const float*)b.put<uint32_t>(cdf, n_lights);
    sc.numLights = n_lights; sc.cdfSum = wordToFloat(cdf[n_lights - 1u]);
    uint32_t* dRows = b.put<uint32_t>(rows43, (size_t)43 * n);
    uint32_t* dD = direct12 ? b.get<uint32_t>((size_t)12 * n) : nullptr;
    uint32_t* dI = indirect10 ? b.get<uint32_t>((size_t)10 * n) : nullptr;
    if (!sc.lights || !sc.cdf || !dRows || (direct12 && !dD) || (indirect10 && !dI)) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    lm_kernel_table()->kat_shade(r->stream, sc, n, W, dRows, fast, dD, dI);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    if (direct12) LM_HIP(hipMemcpy(direct12, dD, (size_t)12 * n * 4, hipMemcpyDeviceToHost));
    if (indirect10) LM_HIP(hipMemcpy(indirect10, dI, (size_t)10 * n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int lumen_mi_test_extract(lumen_mi_renderer* r, uint32_t n, const uint32_t* hits9, const uint32_t* rays9, uint32_t* out35)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                        // before any renderer state is read
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!n || !hits9 || !rays9 || !out35) return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    LM_HIP(hipSetDevice(r->device));
    int rc;
    if ((rc = uploadResources(r))) return rc;
    if ((rc = flatten(r))) return rc;
    LM_HIP(hipStreamSynchronize(r->stream));
    if ((rc = syncScene(r, r->stream))) return rc;
    for (uint32_t i = 0; i < n; i++) if (hits9[9u * i] >= r->entries.size()) return fail(LUMEN_MI_ERR_INVALID, "hit record names a table entry the scene does not have");
    Bufs b;
    uint32_t* dH = b.put<uint32_t>(hits9, (size_t)9 * n); uint32_t* dR = b.put<uint32_t>(rays9, (size_t)9 * n); uint32_t* dO = b.get<uint32_t>((size_t)35 * n);
    if (!dH || !dR || !dO) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    lm_kernel_table()->kat_extract(r->stream, r->dscene, n, dH, dR, dO);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out35, dO, (size_t)35 * n * 4, hipMemcpyDeviceToHost));
    return 0;
}

/* The texture fetch of the extraction kernels (lm_tex2D) on n normalised coordinates of one texture: what tex2D<float4>(PTTexture object, u, v) returns in the reference
 * (PTTexture.cpp:35-74: linear filter, wrap, normalised float read, sRGB decode per texel) under the filter rule in force (tuning key tex_filter, decision D6). */
int lumen_mi_test_tex2d(lumen_mi_renderer* r, lumen_mi_handle texture, uint32_t n, const float* uv2, float* out4)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    size_t id;
    if (!n || !uv2 || !out4 || !unh(texture, H_TEXTURE, r->textures.size(), id)) return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    LM_HIP(hipSetDevice(r->device));
    int rc;
    if ((rc = uploadResources(r))) return rc;
    Bufs b;
    float2* dU = (float2*)b.put<float>(uv2, (size_t)2 * n); float4* dO = (float4*)b.get<float>((size_t)4 * n);
    if (!dU || !dO) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    lm_kernel_table()->kat_tex2d(r->stream, r->dscene, n, (int)id, dU, dO);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(out4, dO, (size_t)16 * n, hipMemcpyDeviceToHost));
    return 0;
}

/* The depth-0 kernel itself (lm_k_extract0: ExtractSurfaceDataGpu + GenerateMotionVector MotionVectors.cu:8-55 + ResolveDirectLightHits GPUShadeDirect.cu:11-40 fused) on given
 * hit records of the renderer's current render resolution: hits9 / dirs3 per pixel of the window in row-major pixel order, the eye, the motion matrix; results: the G-buffer
 * record the kernel stored ([n][8][4] floats as lumen_mi_get_gbuffer gives them), motion vectors (half2 bits) and the DIRECT channel it initialised. */
int lumen_mi_test_extract0(lumen_mi_renderer* r, const uint32_t* hits9, const uint32_t* dirs3, const uint32_t* eye3, const uint32_t* matrix16, float* gbuffer, uint32_t* motion, float* direct)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                        // before any renderer state is read
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!hits9 || !dirs3 || !eye3 || !matrix16 || !gbuffer || !motion || !direct) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    LM_HIP(hipSetDevice(r->device));
    { std::lock_guard<std::mutex> sl(r->settingsMutex); r->settings = r->pending; }
    int rc;
    if ((rc = uploadResources(r))) return rc;
    if ((rc = flatten(r))) return rc;
    LM_HIP(hipStreamSynchronize(r->stream));
    if ((rc = syncScene(r, r->stream))) return rc;
    if ((rc = ensureFrameBuffers(r))) return rc;
    LmFrame fr = r->fr;
    const uint32_t n = fr.n;
    std::vector<float4> rd(n); std::vector<uint4> hh(n);
    for (uint32_t i = 0; i < n; i++) {                            // queue slot i = pixel i here (any permutation of the pixels is a valid queue: the pixel rides in rayD.w)
        const uint32_t* h = hits9 + 9u * i;
        if (h[0] >= r->entries.size()) return fail(LUMEN_MI_ERR_INVALID, "hit record names a table entry the scene does not have");
        float d[3]; memcpy(d, dirs3 + 3u * i, 12);
        float w; const uint32_t li = h[6] * fr.ww + h[5]; memcpy(&w, &li, 4);
        rd[i] = make_float4(d[0], d[1], d[2], w);
        hh[i] = make_uint4(h[0], h[1], (h[2] & 0xffffu) | (h[3] << 16), h[4]);
    }
    fr.motion = r->dMotion[0].p; fr.direct = r->dDirect[0].p; fr.indirect = r->dIndirect[0].p; fr.counters = r->dCounters.p; fr.hits = r->dHits[0].p;
    fr.rayD[0] = r->dRay[1].p; fr.owedSet = -1; fr.deferred = 0;
    LM_HIP(hipMemcpy(fr.rayD[0], rd.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(fr.hits, hh.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    LM_HIP(hipMemset(fr.counters, 0, LM_CNT_WORDS * 4));
    LmCamera cam{};
    for (int k = 0; k < 3; k++) cam.eye[k] = wordToFloat(eye3[k]);
    for (int k = 0; k < 16; k++) cam.prevViewProj[k] = wordToFloat(matrix16[k]);
    lm_kernel_table()->extract0(r->stream, r->gridFor(n, 8), r->dscene, fr, cam, 0, 0u, 0, 1, fr.counters + LM_CNT_RAYS(1));
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    LM_HIP(hipMemcpy(gbuffer, fr.gbuf[0], (size_t)n * 8 * 16, hipMemcpyDeviceToHost));
    LM_HIP(hipMemcpy(motion, fr.motion, (size_t)n * 4, hipMemcpyDeviceToHost));
    LM_HIP(hipMemcpy(direct, fr.direct, (size_t)n * 16, hipMemcpyDeviceToHost));
    // the hook wrote the renderer's LIVE G-buffer, motion, DIRECT and counter buffers: what a traced frame left there (temporal history, a deferred history pass) is gone.
    // The next frame therefore starts as after a resize — buffers cleared, reservoirs reset, frame and blend counters at 0 (ensureFrameBuffers) — instead of reading them.
    r->owed.valid = false; r->allocN = 0;
    return 0;
}

int lumen_mi_test_primary_rays(lumen_mi_renderer* r, uint32_t W, uint32_t H, uint32_t frame_count, const uint32_t* cam_uvw_eye12, uint32_t* out11)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);                                        // before any renderer state is read
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!W || !H || !cam_uvw_eye12 || !out11) return fail(LUMEN_MI_ERR_INVALID, "bad argument");
    LM_HIP(hipSetDevice(r->device));
    const uint32_t n = W * H;
    Bufs b;
    LmFrame fr{};
    fr.W = W; fr.H = H; fr.ww = W; fr.wh = H; fr.tx1 = W; fr.ty1 = H; fr.n = n;
    fr.rayD[0] = b.get<float4>(n); fr.counters = b.get<uint32_t>(LM_CNT_WORDS);
    if (!fr.rayD[0] || !fr.counters) return fail(LUMEN_MI_ERR_DEVICE, "allocation failed");
    LmCamera cam{};
    for (int k = 0; k < 3; k++) { cam.U[k] = wordToFloat(cam_uvw_eye12[k]); cam.V[k] = wordToFloat(cam_uvw_eye12[3 + k]); cam.Wv[k] = wordToFloat(cam_uvw_eye12[6 + k]); cam.eye[k] = wordToFloat(cam_uvw_eye12[9 + k]); }
    lm_kernel_table()->primary(r->stream, r->gridFor(n, 8), fr, cam, frame_count);
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(r->stream));
    std::vector<float4> d(n);
    LM_HIP(hipMemcpy(d.data(), fr.rayD[0], (size_t)n * 16, hipMemcpyDeviceToHost));
    uint32_t cnt = 0;
    LM_HIP(hipMemcpy(&cnt, fr.counters + LM_CNT_RAYS(0), 4, hipMemcpyDeviceToHost));
    if (cnt != n) return fail(LUMEN_MI_ERR_DEVICE, "primary ray count");
    std::vector<uint8_t> seen(n, 0);
    const float one = 1.f;
    for (uint32_t i = 0; i < n; i++) {             // queue slot -> pixel row; origin and contribution are the kernel's implicit contract (the eye; 1, 1, 1): see lm_k_primary
        uint32_t li; memcpy(&li, &d[i].w, 4);
        if (li >= n || seen[li]) return fail(LUMEN_MI_ERR_DEVICE, "primary ray queue is not a permutation of the pixels");
        seen[li] = 1;
        uint32_t* o = out11 + 11u * li;
        o[0] = li % W; o[1] = li / W;
        memcpy(o + 2, cam_uvw_eye12 + 9, 12); memcpy(o + 5, &d[i].x, 12);
        for (int k = 0; k < 3; k++) memcpy(o + 8 + k, &one, 4);
    }
    return 0;
}

}  // extern "C"

#endif   // LUMEN_MI_TEST_HOOKS
