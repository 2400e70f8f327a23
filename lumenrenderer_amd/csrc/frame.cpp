// frame.cpp — frame buffers and the frame graph of one TraceFrame (streams, events, cross-frame pipelining).
// Follows WaveFrontRenderer::TraceFrame (WaveFrontRenderer.cpp:435-1089) for order of operations, seed evolution and counters,
// but enqueues the whole frame without host round trips (the reference synchronises ~40 times per frame).
#include <cmath>
#include "renderer_state.h"

namespace lmr {

int ensureFrameBuffers(R* r)
{
    const uint32_t W = r->settings.render_width, H = r->settings.render_height;
    if (!r->windowSet) { r->wx0 = 0; r->wy0 = 0; r->wx1 = W; r->wy1 = H; }
    if (r->wx1 > W || r->wy1 > H || r->wx0 >= r->wx1 || r->wy0 >= r->wy1) return fail(LUMEN_MI_ERR_INVALID, "render window outside the image");
    const uint32_t ww = r->wx1 - r->wx0, wh = r->wy1 - r->wy0, n = ww * wh;
    if (r->tileSet && (r->ox0 < r->wx0 || r->oy0 < r->wy0 || r->ox1 > r->wx1 || r->oy1 > r->wy1)) return fail(LUMEN_MI_ERR_INVALID, "owned tile outside the render window");
    LmFrame& f = r->fr;
    if (!r->dTotals.p) {
        if (r->dTotals.ensure(LM_CNT_WORDS + 1) || hipMemsetAsync(r->dTotals.p, 0, (LM_CNT_WORDS + 1) * sizeof(unsigned long long), r->stream) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "counter totals allocation failed");
    }
    f.totals = r->dTotals.p;
    // compared against the COMMITTED state: allocN is only set once every allocation and reset below has succeeded
    const bool realloc = n != r->allocN || f.W != W || f.H != H || f.x0 != r->wx0 || f.y0 != r->wy0 || f.ww != ww;
    if (!realloc) {
        f.tx0 = r->tileSet ? r->ox0 - r->wx0 : 0; f.ty0 = r->tileSet ? r->oy0 - r->wy0 : 0; f.tx1 = r->tileSet ? r->ox1 - r->wx0 : ww; f.ty1 = r->tileSet ? r->oy1 - r->wy0 : wh;
        return 0;
    }
    r->allocN = 0;                                        // a failure below leaves "nothing allocated": the next call starts over
    r->owed.valid = false;                                // ... and nothing owed: the deferred history passes would run on the buffers freed below
    int bad = 0;
    for (int i = 0; i < 6; i++) bad |= r->dRay[i].ensure(n) | r->dRay[6 + i].ensure(n) | r->dTailRay[i].ensure(n);
    for (int i = 0; i < 6; i++) bad |= r->dSh[i].ensure(n);
    for (int i = 0; i < 4; i++) bad |= r->dSh2[i].ensure(n);
    for (int i = 0; i < 3; i++) bad |= r->dGbuf[i].ensure((size_t)8 * n) | r->dProbe[i].ensure(n);
    for (int i = 0; i < 2; i++) bad |= r->dMotion[i].ensure(n);
    bad |= r->dReuseMask.ensure(n) | r->dHazard[0].ensure(n) | r->dHazard[1].ensure(n);
    const size_t nTiles = (size_t)((ww + 15u) / 16u) * ((wh + 15u) / 16u);
    for (int i = 0; i < 3; i++) bad |= r->dRareTile[i].ensure(nTiles);
    for (int i = 0; i < 5; i++) bad |= r->dRes[i].ensure((size_t)4 * n) | r->dResC[i].ensure(n);
    for (int i = 0; i < 2; i++) bad |= r->dDirect[i].ensure(n) | r->dIndirect[i].ensure(n);
    bad |= r->dCombined.ensure(n) | r->dHits[0].ensure(n) | r->dHits[1].ensure(n) | r->dOutput.ensure(n);
    bad |= r->dCounters.ensure(2 * LM_CNT_WORDS) | r->dBags.ensure(50 * 1000);
    if (bad) return fail(LUMEN_MI_ERR_DEVICE, "frame buffer allocation failed");
    f.W = W; f.H = H; f.x0 = r->wx0; f.y0 = r->wy0; f.ww = ww; f.wh = wh; f.n = n;
    f.tx0 = r->tileSet ? r->ox0 - r->wx0 : 0; f.ty0 = r->tileSet ? r->oy0 - r->wy0 : 0; f.tx1 = r->tileSet ? r->ox1 - r->wx0 : ww; f.ty1 = r->tileSet ? r->oy1 - r->wy0 : wh;
    for (int q = 0; q < 2; q++) { f.rayO[q] = r->dRay[3 * q].p; f.rayD[q] = r->dRay[3 * q + 1].p; f.rayC[q] = r->dRay[3 * q + 2].p; }
    f.shO = r->dSh[0].p; f.shD = r->dSh[1].p; f.shR = r->dSh[2].p;
    f.visO = r->dSh2[0].p; f.visD = r->dSh2[1].p; f.vis2O = r->dSh2[2].p; f.vis2D = r->dSh2[3].p;
    f.hits = r->dHits[0].p;
    for (int i = 0; i < 3; i++) { f.gbuf[i] = r->dGbuf[i].p; f.probe[i] = r->dProbe[i].p; }
    for (int i = 0; i < 5; i++) { f.res[i] = r->dRes[i].p; f.resC[i] = r->dResC[i].p; }
    f.reuseMask = r->dReuseMask.p;
    for (int i = 0; i < 3; i++) f.rareTile[i] = r->dRareTile[i].p;
    f.motion = r->dMotion[0].p; f.direct = r->dDirect[0].p; f.indirect = r->dIndirect[0].p; f.combined = r->dCombined.p; f.output = r->dOutput.p;
    f.counters = r->dCounters.p; f.bags = r->dBags.p;
    // ResizeBuffers (WaveFrontRenderer.cpp:1424-1540): history is dropped; reservoirs reset (ReSTIRKernels.cu:36-47)
    hipStream_t st = r->stream;
    for (int i = 0; i < 3; i++) if (hipMemsetAsync(f.gbuf[i], 0, (size_t)8 * n * sizeof(float4), st) != hipSuccess || hipMemsetAsync(f.probe[i], 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    for (int i = 0; i < 5; i++) if (hipMemsetAsync(f.res[i], 0, (size_t)4 * n * sizeof(float4), st) != hipSuccess || hipMemsetAsync(f.resC[i], 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    if (hipMemsetAsync(f.combined, 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    for (int i = 0; i < 3; i++) if (hipMemsetAsync(f.rareTile[i], 0, nTiles * sizeof(uint32_t), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    if (hipMemsetAsync(f.output, 0, (size_t)n * sizeof(uchar4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    r->fenceNeeded = 2;
    r->haveEst = false; r->cntPending[0] = r->cntPending[1] = false;
    if (r->dSwap.ensure(16) || hipMemsetAsync(r->dSwap.p, 0, 16 * sizeof(int), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "swap index allocation failed");
    f.swap = r->dSwap.p;
    f.deferred = 0; f.owedSet = -1; f.hazardList = nullptr;
    r->owed.valid = false;                                // the reservoirs were reset: nothing is owed
    r->blendCounter = 0; r->frameIndex = 0; r->gbufIndex = 0; r->lastGbuf = 0;
    r->allocN = n;
    return 0;
}

void invert4(const float* m, float* out)
{
    double a[16], inv[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    inv[0] = a[5]*a[10]*a[15] - a[5]*a[11]*a[14] - a[9]*a[6]*a[15] + a[9]*a[7]*a[14] + a[13]*a[6]*a[11] - a[13]*a[7]*a[10];
    inv[4] = -a[4]*a[10]*a[15] + a[4]*a[11]*a[14] + a[8]*a[6]*a[15] - a[8]*a[7]*a[14] - a[12]*a[6]*a[11] + a[12]*a[7]*a[10];
    inv[8] = a[4]*a[9]*a[15] - a[4]*a[11]*a[13] - a[8]*a[5]*a[15] + a[8]*a[7]*a[13] + a[12]*a[5]*a[11] - a[12]*a[7]*a[9];
    inv[12] = -a[4]*a[9]*a[14] + a[4]*a[10]*a[13] + a[8]*a[5]*a[14] - a[8]*a[6]*a[13] - a[12]*a[5]*a[10] + a[12]*a[6]*a[9];
    inv[1] = -a[1]*a[10]*a[15] + a[1]*a[11]*a[14] + a[9]*a[2]*a[15] - a[9]*a[3]*a[14] - a[13]*a[2]*a[11] + a[13]*a[3]*a[10];
    inv[5] = a[0]*a[10]*a[15] - a[0]*a[11]*a[14] - a[8]*a[2]*a[15] + a[8]*a[3]*a[14] + a[12]*a[2]*a[11] - a[12]*a[3]*a[10];
    inv[9] = -a[0]*a[9]*a[15] + a[0]*a[11]*a[13] + a[8]*a[1]*a[15] - a[8]*a[3]*a[13] - a[12]*a[1]*a[11] + a[12]*a[3]*a[9];
    inv[13] = a[0]*a[9]*a[14] - a[0]*a[10]*a[13] - a[8]*a[1]*a[14] + a[8]*a[2]*a[13] + a[12]*a[1]*a[10] - a[12]*a[2]*a[9];
    inv[2] = a[1]*a[6]*a[15] - a[1]*a[7]*a[14] - a[5]*a[2]*a[15] + a[5]*a[3]*a[14] + a[13]*a[2]*a[7] - a[13]*a[3]*a[6];
    inv[6] = -a[0]*a[6]*a[15] + a[0]*a[7]*a[14] + a[4]*a[2]*a[15] - a[4]*a[3]*a[14] - a[12]*a[2]*a[7] + a[12]*a[3]*a[6];
    inv[10] = a[0]*a[5]*a[15] - a[0]*a[7]*a[13] - a[4]*a[1]*a[15] + a[4]*a[3]*a[13] + a[12]*a[1]*a[7] - a[12]*a[3]*a[5];
    inv[14] = -a[0]*a[5]*a[14] + a[0]*a[6]*a[13] + a[4]*a[1]*a[14] - a[4]*a[2]*a[13] - a[12]*a[1]*a[6] + a[12]*a[2]*a[5];
    inv[3] = -a[1]*a[6]*a[11] + a[1]*a[7]*a[10] + a[5]*a[2]*a[11] - a[5]*a[3]*a[10] - a[9]*a[2]*a[7] + a[9]*a[3]*a[6];
    inv[7] = a[0]*a[6]*a[11] - a[0]*a[7]*a[10] - a[4]*a[2]*a[11] + a[4]*a[3]*a[10] + a[8]*a[2]*a[7] - a[8]*a[3]*a[6];
    inv[11] = -a[0]*a[5]*a[11] + a[0]*a[7]*a[9] + a[4]*a[1]*a[11] - a[4]*a[3]*a[9] - a[8]*a[1]*a[7] + a[8]*a[3]*a[5];
    inv[15] = a[0]*a[5]*a[10] - a[0]*a[6]*a[9] - a[4]*a[1]*a[10] + a[4]*a[2]*a[9] + a[8]*a[1]*a[6] - a[8]*a[2]*a[5];
    const double det = a[0]*inv[0] + a[1]*inv[4] + a[2]*inv[8] + a[3]*inv[12];
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] / det);
}

// timing helpers: HIP events on the renderer's own stream
void evBegin(R* r, int cls, size_t& slot)
{
    slot = (size_t)-1;
    if (!r->timing || (r->timing == 2 && cls != 0 && cls != 4)) return;
    if (r->evUsed == r->evPool.size()) { R::EvPair p; if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; p.cls = 0; r->evPool.push_back(p); }
    slot = r->evUsed++;
    r->evPool[slot].cls = cls;
    (void)hipEventRecord(r->evPool[slot].a, r->stream);
}
void evEnd(R* r, size_t slot) { if (slot != (size_t)-1) (void)hipEventRecord(r->evPool[slot].b, r->stream); }
void evBegin2(R* r, int cls, size_t& slot, hipStream_t s)
{
    slot = (size_t)-1;
    if (!r->timing || (r->timing == 2 && cls != 0 && cls != 4)) return;
    if (r->evUsed == r->evPool.size()) { R::EvPair p; if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; p.cls = 0; r->evPool.push_back(p); }
    slot = r->evUsed++;
    r->evPool[slot].cls = cls;
    (void)hipEventRecord(r->evPool[slot].a, s);
}
void evEnd2(R* r, size_t slot, hipStream_t s) { if (slot != (size_t)-1) (void)hipEventRecord(r->evPool[slot].b, s); }

// Camera::GetVectorData (Camera.cpp:79-93,122-128): image-plane half sizes from the vertical field of view, focal length 1
void cameraVectors(const float* right, const float* up, const float* fwd, float fovY, float aspect, float* U, float* V, float* Wv)
{
    const float halfY = 1.0f * (float)tan((double)(fovY * 0.01745329251994329576923690768489f) * 0.5);
    const float halfX = halfY * aspect;
    for (int k = 0; k < 3; k++) { U[k] = right[k] * halfX; V[k] = up[k] * halfY; Wv[k] = fwd[k] * 1.0f; }
}
// M = projection(fovY, aspect, 0.5, 10000) * inverse(previous camera world matrix), row major   (WaveFrontRenderer.cpp:763-776, CPUShadingKernels.cu:39)
void motionMatrix(const float* prevCamWorld, float fovY, float aspect, float* M)
{
    float proj[16] = {0}, invPrev[16];
    const float tanHalf = (float)tan((double)(fovY * 0.01745329251994329576923690768489f) / 2.0);
    const float zn = 0.5f, zf = 10000.f;
    proj[0] = 1.0f / (aspect * tanHalf); proj[5] = 1.0f / tanHalf;
    proj[10] = -(zf + zn) / (zf - zn); proj[11] = -(2.0f * zf * zn) / (zf - zn); proj[14] = -1.0f;
    invert4(prevCamWorld, invPrev);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0.f; for (int k = 0; k < 4; k++) s += proj[i * 4 + k] * invPrev[k * 4 + j]; M[i * 4 + j] = s; }
}

// The history-building passes of the last frame (ReSTIR.cpp:181-233: both spatial passes, CombineReservoirBuffers), with that frame's parameters; the
// kernels return at once unless the swap chain has turned since (kernels.hip lm_reuse_owed).  mode 1: inside the next frame, 2: between frames.
void launchOwedReuse(R* r, hipStream_t s, int mode)
{
    const R::OwedReuse& o = r->owed;
    LmFrame fo = o.fr;
    fo.swap = r->fr.swap; fo.deferred = mode;
    const LmKernelTable* K = r->K;
    K->spatial(s, o.tiles, fo, o.gbuf, LM_RES_OWED, 2, o.seed, 30, 0, o.fast);
    K->spatial(s, o.tiles, fo, o.gbuf, 2, 3, o.seed, 0, 1, o.fast);
    K->combine(s, o.tiles, fo, o.gbuf, LM_RES_OWED, 3, wangHash(o.seed), o.fast);
}

int traceFrameAsync(R* r)
{
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "lumen_mi_init has not been called");
    { std::lock_guard<std::mutex> lk(r->settingsMutex); r->settings = r->pending; }          // WaveFrontRenderer.cpp:480-505
    if (hipSetDevice(r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    int rc;
    if ((rc = uploadResources(r))) return rc;
    if ((rc = flatten(r))) return rc;
    if ((rc = buildLights(r))) return rc;                                                     // :456
    r->countersValid = false;
    if (r->totalEmissive == 0 || r->lights.empty()) return LUMEN_MI_NO_LIGHTS;                // :459-464
    {   // scene edits since the last frame go to the device on the first wave stream, behind the merge of the frame two back (the last
        // reader of the scene set that is rewritten); see SceneSet.  With two wave streams the refit kernels of consecutive frames still
        // share one stream (and their scratch buffers); an odd frame's own stream waits for them below (evScene).
        hipStream_t su = (r->overlap && r->aux != nullptr) ? r->aux : r->stream;
        if (su != r->stream) LM_HIP(hipStreamWaitEvent(su, r->evMerge[r->framePar], 0));
        if ((rc = syncScene(r, su))) return rc;
        if (su != r->stream && r->waveStreams == 2 && r->framePar) { LM_HIP(hipEventRecord(r->evScene, su)); LM_HIP(hipStreamWaitEvent(r->aux2, r->evScene, 0)); }
    }
    if ((rc = ensureFrameBuffers(r))) return rc;
    const LmKernelTable* K = r->K;
    // schedule fuzzing: per frame a random subset of the four streams is slow; three quarters of the launches on a slow stream are
    // preceded by an idle wavefront of 0 - 1.5 ms (the order of magnitude of the kernels themselves); see lm_k_spin
    auto fuzzNext = [&]() { uint32_t x = r->fuzz; x ^= x << 13; x ^= x >> 17; x ^= x << 5; r->fuzz = x ? x : 1u; return x; };
    const uint32_t slowStreams = r->fuzz ? (fuzzNext() >> 7) & 15u : 0u;
    auto Z = [&](hipStream_t s) {
        if (!r->fuzz) return;
        const int which = s == r->stream ? 0 : s == r->aux ? 1 : s == r->aux2 ? 2 : 3;
        const uint32_t x = fuzzNext();
        if (((slowStreams >> which) & 1u) && (x & 3u) != 0u) K->spin(s, (x >> 8) % 150000u);
    };
    hipStream_t st = r->stream;
    LmFrame& fr = r->fr;
    const uint32_t depthMax = std::min<uint32_t>(r->settings.depth, LM_MAX_DEPTH);
    // "current" / "previous" surface data (the reference toggles two buffers, WaveFrontRenderer.cpp:1045-1049); here three physical
    // sets rotate, so that the next frame's extraction does not wait for this frame's temporal pass
    const int currentIndex = r->gbufIndex, temporalIndex = (r->gbufIndex + 2) % 3;
    const bool blend = r->settings.blend_output != 0;
    // 0 exact; 1 fast, common launch only (no material can produce a surface outside the contracted evaluation); 2 fast, common + rare launch
    const int fastRs = r->fastResample ? (r->anyRareMaterial ? 2 : 1) : 0;
    // visibility rays of the ReSTIR passes: packets under the same rule (tuning key packet_visibility: 1 on, 0 off, -1 automatic)
    const bool visPackets = r->packetVisibility > 0 || (r->packetVisibility < 0 && (uint64_t)r->triEntry.size() * 4u < (uint64_t)r->fr.n && r->fr.n > 0);
    const bool usePackets = r->packetPrimary > 0 || (r->packetPrimary < 0 && (uint64_t)r->triEntry.size() * 3u < (uint64_t)r->fr.n && r->fr.n > 0);

    // camera (Camera.cpp:79-93,122-140; aspect = render W/H, WaveFrontRenderer.cpp:577)
    LmCamera cam;
    const float aspect = (float)fr.W / (float)fr.H;
    for (int k = 0; k < 3; k++) cam.eye[k] = r->camPos[k];
    cameraVectors(r->camRight, r->camUp, r->camFwd, r->fovY, aspect, cam.U, cam.V, cam.Wv);
    float camWorld[16] = {r->camRight[0], r->camUp[0], r->camFwd[0], r->camPos[0], r->camRight[1], r->camUp[1], r->camFwd[1], r->camPos[1],
                          r->camRight[2], r->camUp[2], r->camFwd[2], r->camPos[2], 0, 0, 0, 1};
    if (!r->havePrev) { memcpy(r->prevCamWorld, camWorld, sizeof camWorld); r->havePrev = true; }
    motionMatrix(r->prevCamWorld, r->fovY, aspect, cam.prevViewProj);

    // ---- frame graph.  Streams: main `st` (ReSTIR chain, merge), `sx` (frame front + indirect waves), aux2 (NEE shadow
    // rays), aux3 (second ReSTIR visibility pass).  Frames are software-pipelined: the front of frame i+1 (primary rays,
    // first closest-hit launch, surface extraction, first continuation) is queued on `sx` behind the waves of frame i and
    // runs beside the ReSTIR tail of frame i on `st`.  What that needs: DIRECT / INDIRECT and the counter block are
    // double-buffered by frame parity; extraction waits for frame i's temporal pass (the last reader of the G-buffer /
    // probe plane / motion vectors it overwrites); a frame's front waits for the merge of the frame two back (owner of
    // the same parity buffers).  Accumulation order per pixel is unchanged, so results equal the serial order bit for bit.
    const bool overlap = r->overlap && r->aux != nullptr;
    const int par = r->framePar; r->framePar ^= 1;
    // Two wave streams (`wave_streams` 2; default 1): the path-tracing launches of a frame — closest hit, extraction / shading, NEE shadow rays,
    // path tail — run in series on ONE stream, and even / odd frames use different ones, so the wave chains of consecutive frames (which do
    // not depend on each other) overlap instead of queueing behind each other; ray queues, shadow queue and hit records exist once per
    // parity for that.  (One wave stream: every frame's waves on `aux`, shadows and tail beside them on `aux2`.)
    const bool twoWave = overlap && r->waveStreams == 2 && r->sortRays == 0;
    hipStream_t sx = overlap ? ((twoWave && par) ? r->aux2 : r->aux) : st;
    {
        const int b = twoWave ? par : 0;
        for (int q2 = 0; q2 < 2; q2++) { fr.rayO[q2] = r->dRay[6 * b + 3 * q2].p; fr.rayD[q2] = r->dRay[6 * b + 3 * q2 + 1].p; fr.rayC[q2] = r->dRay[6 * b + 3 * q2 + 2].p; }
        fr.shO = r->dSh[3 * b].p; fr.shD = r->dSh[3 * b + 1].p; fr.shR = r->dSh[3 * b + 2].p;
        fr.hits = r->dHits[b].p;
    }
    fr.deferred = 0;
    fr.owedSet = r->owed.valid ? r->owed.gbuf : -1; fr.hazardList = r->dHazard[par].p;      // lazy reuse: the extraction lists the entries that outlive this frame's pick
    fr.motion = r->dMotion[par].p;
    fr.direct = r->dDirect[par].p; fr.indirect = r->dIndirect[par].p; fr.counters = r->dCounters.p + (size_t)par * LM_CNT_WORDS;
    if (overlap) {
        if (r->fenceNeeded) { LM_HIP(hipEventRecord(r->evTop, st)); LM_HIP(hipStreamWaitEvent(sx, r->evTop, 0)); }
        LM_HIP(hipStreamWaitEvent(sx, r->evMerge[par], 0));
    }
    if (r->fenceNeeded > 0) --r->fenceNeeded;
    size_t evAll; evBegin2(r, 4, evAll, sx);
#if !LM_PRIMARY_CLEARS
    LM_HIP(hipMemsetAsync(fr.counters, 0, LM_CNT_WORDS * sizeof(uint32_t), sx));
#endif
    // the map of tiles that need the exact launch of the fast ReSTIR passes, of the G-buffer set this frame's extraction fills (its last readers belonged to the frame
    // three back, which the wait for evMerge above has seen end).  Only scenes with a material outside the contracted evaluation ever read it.
    // The extraction (lm_k_extract0) marks tiles only while the map can be read: the frame's view of the pointers is the buffers in the fast mode with such a material
    // (fastRs == 2) and NULL otherwise, so exact-mode frames of a glass-heavy scene no longer pay same-address stores for nothing.  Invariant of the map: a
    // SUPERSET — a set bit only ever makes a block of the second launch look at its pixels; a frame that switches to fastRs == 2 clears its set right here first.
    for (int i = 0; i < 3; i++) fr.rareTile[i] = fastRs == 2 ? r->dRareTile[i].p : nullptr;
    if (fastRs == 2) LM_HIP(hipMemsetAsync(fr.rareTile[currentIndex], 0, (size_t)((fr.ww + 15u) / 16u) * ((fr.wh + 15u) / 16u) * sizeof(uint32_t), sx));
    if (!blend) { Z(st); K->clear(st, r->gridFor(fr.n, 8), fr.combined, fr.n); }                        // :559
    ++r->frameCount;                                                                          // :593
    // the packet kernel of the primary wave generates its rays itself (kernels.hip lm_k_trace_primary_packet); the counting build traces per lane, from the plane
    const bool fusePrimary = usePackets && !r->instrumented && r->fusePrimary != 0;
    if (!fusePrimary) { Z(sx); K->primary(sx, r->gridFor(fr.n, 8), fr, cam, r->frameCount); }
    const uint32_t frameCountPrimary = r->frameCount;
    uint32_t seed = wangHash(r->frameCount);                                                  // :685
    LmScene scx = r->dscene;                                                                  // same scene, its own stack-spill area
    if (overlap) scx.spill += (size_t)((twoWave && par) ? 2 : 1) * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
    // Persistent traversal grids, in blocks per CU.  Eight fill every wave slot of the chip with ONE kernel, and a persistent kernel frees its slots only when it ends — the blocks
    // of the history passes and of the candidate pick beside it wait for that.  Under overlap two launches therefore take HALF the grid when that is their own loss to bear
    // (profiles/r06_trace_blocks_ab.txt, interleaved on one box, every workload; results never depend on a grid):
    //   * the two visibility passes, when the light list is short (<= 64 records: their rays all aim at the same few lights, short coherent any-hit walks; +0.5 ... 3 % in every
    //     mode on C2 / C2t / C4 / C5 / Sandbox; with hundreds of lights the passes are long and the lazy frame waits for them: C3 -3.4 %, LowpolyRoom -2.4 %);
    //   * the primary-ray launch, when the history passes run in this frame (eager / exact: the ReSTIR stream is the critical chain and runs beside it; lazy frames hang on the
    //     wave stream instead: C4 -2 %), the window is large (>= 2 Mpixel) and the tree small enough to stay in cache (<= 2 M triangles; C5's 10 M need every slot to hide HBM
    //     latency: -3.7 %): C2 +0.5 % fast / +1.1 % exact on top, C4 +3.3 % / +3.0 %.
    // traceBlocksMain / traceBlocksVis > 0 (LUMEN_MI_TRACE_BLOCKS_MAIN / _VIS) force a size.  One stream: the full grid for every launch.
    const bool lazyFrame = r->lazyReuse > 0 || (r->lazyReuse < 0 && (depthMax & 1u) == 0u);      // tuning key lazy_reuse; -1: at even path depths
    const int autoMain = (!lazyFrame && fr.n >= (1u << 21) && r->triEntry.size() <= 2000000u) ? 4 : 8;
    const int autoVis = r->dscene.numLights <= 64u ? 4 : 8;
    const int gridAux = r->numCU * r->traceBlocksAux;
    const int gridMain = overlap ? r->numCU * (r->traceBlocksMain > 0 ? r->traceBlocksMain : autoMain) : gridAux;
    const int gridVis = overlap ? r->numCU * (r->traceBlocksVis > 0 ? r->traceBlocksVis : autoVis) : gridAux;
    const int tiles = (int)(((fr.ww + 15u) / 16u) * ((fr.wh + 15u) / 16u));
    // Deep waves hold too few rays to fill the machine; from the first wave expected to be shorter than `tailBelow` rays the
    // remaining depths run as one launch.  The expectation comes from the counters of the most recent frame whose
    // asynchronous read-back has already landed (no host synchronisation; any choice gives the same image).
    for (int p : {par ^ 1, par}) {
        if (r->cntPending[p] && hipEventQuery(r->evCnt[p]) == hipSuccess) {
            for (uint32_t dd = 0; dd <= LM_MAX_DEPTH; dd++) r->estRays[dd] = r->pinnedCounters[p][LM_CNT_RAYS(dd)];
            r->haveEst = true; r->cntPending[p] = false;
            break;
        }
    }
    int tailDepth = (int)depthMax;
    // (fast ReSTIR mode shortens the candidate / reuse kernels: the wave chain is then the critical path at every window size, and the
    // larger threshold measured +4.6 % on C2)
    // round 3 (the temporal pass got cheaper, the wave chain is the longer one by more): 100 000 in fast mode on large windows, i.e. on C2 the wave of 92 k rays joins the
    // tail: +1.8 % (C2), +4.2 % (textured C2), +-0 (C4, C5), -0.9 % (C3), five / three interleaved runs each on one box; exact mode keeps 16 384 (65 536: -1.2 %,
    // 120 000: -2.9 %) — profiles/r03_knobs_ab.txt
    // round 5: on small windows the threshold also follows the TREE.  A wave that stays on the queue kernels costs the critical chain three launches of about T0 each, and
    // T0 is the dependent chain of the longest ray — it grows with the depth of the tree; what it saves is the time the same paths would spend in the tail at ~9 of 64
    // lanes.  720p, fast mode, interleaved on one box (profiles/r05_tail_threshold_ab.txt): the stand-in atrium (262 k triangles; waves 198 k / 54 k / 19 k) loses 11 % when
    // the 54 k wave leaves the tail (threshold 32 768), the reference's default model (LowpolyRoom, 20.5 k triangles; waves 297 k / 55 k / 10 k) GAINS 4.3 % from exactly
    // that.  Rule: 65 536 scaled by (log2(triangles) - 10) / 8, clamped to [1/4, 1] — 1 for the atrium (35 k would be wrong there), 0.54 for the room (35 k: the 55 k
    // wave runs on the queue kernels, the 10 k wave joins the tail).
    const double lt = std::log2((double)std::max<size_t>(2, r->triEntry.size()));
    // Exact mode: the ReSTIR chain is the longer one and the wave chain has slack, as on large windows — half the threshold (the atrium at 720p, exact: 32 768 against
    // 65 536 = +8.0 %, the 54 k wave stays on the queue kernels; the room is indifferent between 16 384 and 32 768).
    const uint32_t smallWindow = (uint32_t)((r->fastResample ? 65536.0 : 32768.0) * std::min(1.0, std::max(0.25, (lt - 10.0) / 8.0)));
    const uint32_t tailBelow = r->tailBelow >= 0 ? (uint32_t)r->tailBelow : (fr.n < (1u << 20) ? smallWindow : r->fastResample ? 100000u : 16384u);
    if (tailBelow && r->haveEst) for (uint32_t dd = 1; dd < depthMax; dd++) if (r->estRays[dd] < tailBelow) { tailDepth = (int)dd; break; }
    int q = 0;
    size_t ev;
    bool tailLaunched = false, lazy = false;
    // the queue the path tail reads is its own (double-buffered by frame parity), so that the tail can run on the shadow
    // stream while the wave stream already enqueues the next frame's front into the regular ray queues
    auto withTailQueue = [&](LmFrame f, int queue) {
        f.rayO[queue] = r->dTailRay[3 * par].p; f.rayD[queue] = r->dTailRay[3 * par + 1].p; f.rayC[queue] = r->dTailRay[3 * par + 2].p;
        return f;
    };
    for (uint32_t depth = 0; depth < depthMax; ++depth) {
        uint32_t* inCount = fr.counters + LM_CNT_RAYS(depth);
        uint32_t* outCount = fr.counters + LM_CNT_RAYS(depth + 1);
        const uint32_t seed2 = wangHash(seed);                                                // CPUShadingKernels.cu:178
        const int doIndirect = depth < depthMax - 1 ? 1 : 0;
        if (depth == 0) {
            evBegin2(r, 0, ev, sx);
            if (fusePrimary) { Z(sx); K->trace_primary(sx, gridMain, scx, fr, cam, frameCountPrimary, fr.hits, 0.01f, 5000.f); }
            else { Z(sx); K->trace_closest(sx, gridMain, scx, nullptr, fr.rayD[q], inCount, fr.hits, 0.01f, 5000.f, fr.counters, usePackets ? -1 : r->refillPrimary, cam.eye); }    // :678,:703; primary rays start at the eye
            evEnd2(r, ev, sx);
            if (overlap) LM_HIP(hipStreamWaitEvent(sx, r->evTemporal[par], 0));          // the temporal pass two frames back has read what extraction overwrites
            if (twoWave && r->owed.valid) LM_HIP(hipStreamWaitEvent(sx, r->evFront, 0));     // (that list compares with the previous frame's probe plane, written on the other wave stream)
            evBegin2(r, 2, ev, sx);
            Z(sx); K->extract0(sx, r->gridFor(fr.n, 8), r->dscene, (int)depth + 1 == tailDepth ? withTailQueue(fr, q ^ 1) : fr, cam, currentIndex, seed2, doIndirect, q ^ 1, outCount);   // + depth-0 continuation
            evEnd2(r, ev, sx);
            // the indirect waves follow on the same stream beside ReSTIR on the main stream: both depend only on the G-buffer
            // ReSTIR::Run (Framework/ReSTIR.cpp:65-233) on the main stream.  Candidate generation and the first visibility pass
            // only need this frame's G-buffer; for small windows (multi-GPU tiles, where the dependency chain and not the
            // machine's throughput bounds the frame) they run on their own stream into the fresh-candidate buffer [4], beside
            // the previous frame's spatial passes; the temporal pass picks them up from there.
            const bool pickAhead = overlap && (r->pickAhead >= 0 ? r->pickAhead != 0 : fr.n < (1u << 20));
            // (four streams in total: HIP multiplexes streams onto 4 hardware queues, and a fifth stream cost 11-18 % through false
            // serialisation in every variant tried, also with GPU_MAX_HW_QUEUES=8)
            hipStream_t sp = pickAhead ? r->aux3 : st;
            if (overlap) { LM_HIP(hipEventRecord(r->evFront, sx)); LM_HIP(hipStreamWaitEvent(sp, r->evFront, 0)); }
            // the fresh-candidate buffer is single: the PREVIOUS frame's temporal pass must have consumed it before this frame's
            // candidates overwrite it (the front no longer waits for that pass since the G-buffer rotates through three sets)
            if (pickAhead) LM_HIP(hipStreamWaitEvent(sp, r->evTemporal[par ^ 1], 0));
            // Lazy reuse: what the previous frame left pending comes first — before this frame's candidates overwrite the buffer it completes.  The device
            // decides (kernels.hip lm_reuse_owed): if the swap chain has turned, that frame's history passes run now; if not, they are dropped and only the
            // entries that outlive this frame's pick (pixels flagged in THIS frame: the extraction above has said which) get the count the combine would
            // have left.  At an even path depth the reference computes a history that its own swap quirk never reads.
            if (r->owed.valid) {
                if (overlap && sp != st) LM_HIP(hipStreamWaitEvent(st, r->evFront, 0));
                size_t evOwed; evBegin(r, 3, evOwed);          // class 3 (ReSTIR): the deferred passes are ReSTIR time whenever they really run (odd depth, or waves running out)
                Z(st); launchOwedReuse(r, st, 1);
                if (r->lazyReuse != 2)     // (2: without — wrong on purpose, for the test that shows the completion is observable)
                { LmFrame fo = r->owed.fr; fo.swap = fr.swap; Z(st); K->reuse_counts(st, fo, r->owed.gbuf, fr.hazardList, fr.counters + LM_CNT_HAZARD, r->owed.seed); }
                evEnd(r, evOwed);
                r->owed.valid = false;
            }
            evBegin2(r, 3, ev, sp);
            const int cur = LM_RES_CUR, tmp = LM_RES_PREV, fresh = pickAhead ? 4 : LM_RES_CUR;
            uint32_t rs = wangHash(seed);
            Z(sp); K->fill_bags(sp, r->dscene, fr, seed, 50u * 1000u);
            rs = wangHash(rs);
            const uint32_t tx0 = fr.x0 / 16u, ty0 = fr.y0 / 16u;
            const uint32_t wtx = (fr.x0 + fr.ww + 15u) / 16u - tx0, wty = (fr.y0 + fr.wh + 15u) / 16u - ty0;
            Z(sp); K->pick_primary(sp, (int)(wtx * wty), r->dscene, fr, currentIndex, fresh, rs, fr.counters + LM_CNT_RESTIR(0), fastRs | ((r->pickWide + 1) << 6));   // + visibility rays, pass 1
            LmScene scp = r->dscene;                                 // the pick-ahead stream traces with its own stack-spill area
            if (sp != st) scp.spill += (size_t)3 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
            Z(sp); K->trace_shade(sp, gridVis, scp, fr, fresh, fr.counters + LM_CNT_RESTIR(0), visPackets ? -1 : r->refillVisibility, 0);
            evEnd2(r, ev, sp);
            if (pickAhead) { LM_HIP(hipEventRecord(r->evPick, sp)); LM_HIP(hipStreamWaitEvent(st, r->evPick, 0)); }
            evBegin(r, 3, ev);
            rs = wangHash(rs);
            // lazy reuse (tuning key lazy_reuse; -1: at even path depths): this frame's history passes are left to the next frame (above)
            lazy = lazyFrame;
            Z(st); K->temporal(st, tiles, fr, currentIndex, temporalIndex, cur, tmp, fresh, rs, fr.counters + LM_CNT_RESTIR(1), fastRs);         // + visibility rays, pass 2
            if (overlap) LM_HIP(hipEventRecord(r->evTemporal[par], st));
            rs = wangHash(rs);
            if (!lazy) { Z(st); K->spatial(st, tiles, fr, currentIndex, cur, 2, rs, 30, 0, fastRs | ((fastRs && r->spatialLds) ? (r->spatialLds == 2 ? 32 : 16) : 0)); }
            // second visibility pass (ReSTIR.cpp:211-212) works on the CURRENT buffer, which the second spatial pass does not
            // touch: trace it beside that pass.  (It must follow the first spatial pass, which reads the current buffer.)
            hipStream_t sv = (overlap && !pickAhead) ? r->aux3 : st;
            LmScene scv = r->dscene;
            if (sv != st) {
                scv.spill += (size_t)3 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
                LM_HIP(hipEventRecord(r->evVis, st)); LM_HIP(hipStreamWaitEvent(sv, r->evVis, 0));
            }
            Z(sv); K->trace_shade(sv, gridVis, scv, fr, cur, fr.counters + LM_CNT_RESTIR(1), visPackets ? -1 : r->refillVisibility, lazy ? 2 : 1);   // 2: parks the weights it zeroes for the deferred first spatial pass
            if (sv != st) LM_HIP(hipEventRecord(r->evVisDone, sv));
            // the second spatial pass ends with the combine (one full-screen launch and one read of its own output less) when the visibility pass ran in line before it and no
            // surface needs the second launch's role (exact mode, or a scene without such a material): tuning key fuse_combine
            const bool fuse = !lazy && sv == st && fastRs <= 1 && r->fuseCombine != 0;
            if (fuse) { LmFrame ff = fr; ff.fuseRc = cur; ff.fuseSeed = wangHash(rs); Z(st); K->spatial(st, tiles, ff, currentIndex, 2, 3, rs, 0, 1, fastRs | 64); }
            else if (!lazy) { Z(st); K->spatial(st, tiles, fr, currentIndex, 2, 3, rs, 0, 1, fastRs); }      // the same seed as the first pass (ReSTIR.cpp: one seed for both): same candidates, same verdicts
            if (sv != st) LM_HIP(hipStreamWaitEvent(st, r->evVisDone, 0));
            if (!lazy && !fuse) { Z(st); K->combine(st, tiles, fr, currentIndex, cur, 3, wangHash(rs), fastRs); }
            else { r->owed.valid = true; r->owed.fr = fr; r->owed.gbuf = currentIndex; r->owed.seed = rs; r->owed.fast = fastRs; r->owed.tiles = tiles; }
            evEnd(r, ev);
        } else if ((int)depth >= tailDepth) {
            // path tail: the remaining waves in one launch (kernels.hip lm_k_path_tail) on the shadow stream: its INDIRECT adds
            // follow the previous wave's NEE adds by stream order, and the wave stream is free for the next frame's front
            hipStream_t stl = (overlap && !twoWave) ? r->aux2 : sx;
            LmScene sct = scx;
            if (overlap) {
                if (stl != sx) sct.spill += (size_t)r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
                LM_HIP(hipEventRecord(r->evShade[depth], sx)); LM_HIP(hipStreamWaitEvent(stl, r->evShade[depth], 0));     // the queue's producer is done
            }
            // paths per wavefront; negative = pair mode (kernels.hip): the NEE shadow ray of depth d is traced by a partner lane beside the
            // path's closest-hit query of depth d + 1 (at most 32 paths per wavefront then)
            const int tailL = r->tailLanes > 0 ? r->tailLanes : (fr.n >= 786432u ? 64 : 16);
            // (automatic: on small windows, where the tail is the longest launch of the frame — 1/4 tile of 1440p +4.8 %, 1/8 tile +1.6 %; off from
            // 1.5 Mpixel, where it hides behind the other streams and 64 paths per wavefront cost fewer issue slots: profiles/r03_tail_pair_ab.txt)
            const bool pair = r->tailPair > 0 || (r->tailPair < 0 && fr.n < 1500000u);
            const int tailShape = r->tailRepack > 0 ? 1000 : pair ? -std::min(32, tailL) : tailL;      // 1000: the repacking variant (tuning key tail_repack, kernels.hip)
            evBegin2(r, 5, ev, stl);
            Z(stl); K->path_tail(stl, r->numCU * r->tailGrid, sct, withTailQueue(fr, q), q | (r->fastShade ? 2 : 0), inCount, (int)depth, (int)depthMax, seed, tailShape);
            evEnd2(r, ev, stl);
            if (overlap) LM_HIP(hipEventRecord(r->evTail, stl));
            tailLaunched = true;
            break;
        } else {
            uint32_t* shCount = fr.counters + LM_CNT_SHADOW(depth);
            if (r->sortRays > 0 && (int)depth <= r->sortRays) {
                // reorder this wave's rays into the other queue (free: its rays were consumed by the previous shading launch on this stream)
                if (!r->dSortBins.p) { if (r->dSortBins.ensure(2 * 4096)) return fail(LUMEN_MI_ERR_DEVICE, "sort bins allocation failed"); LM_HIP(hipMemsetAsync(r->dSortBins.p, 0, 2 * 4096 * sizeof(uint32_t), sx)); }
                evBegin2(r, 2, ev, sx);
                Z(sx); K->sort_rays(sx, r->gridFor(fr.n / 4u, 8), r->dscene, fr.rayO[q], fr.rayD[q], fr.rayC[q], fr.rayO[q ^ 1], fr.rayD[q ^ 1], fr.rayC[q ^ 1], inCount, r->dSortBins.p);
                evEnd2(r, ev, sx);
                q ^= 1;
            }
            evBegin2(r, 0, ev, sx);
            Z(sx); K->trace_closest(sx, gridAux, scx, fr.rayO[q], fr.rayD[q], inCount, fr.hits, 0.01f, 5000.f, fr.counters, r->refillBelow, nullptr);
            evEnd2(r, ev, sx);
            if (overlap && !twoWave) LM_HIP(hipStreamWaitEvent(sx, r->evJoin2, 0));       // previous wave's (or frame's) shadow rays consumed (two wave streams: the
                                                                                         // shadow launches are on this stream, and the event is not per frame)
            evBegin2(r, 2, ev, sx);
            Z(sx); K->shade_wave(sx, r->numCU * 8, scx, (int)depth + 1 == tailDepth ? withTailQueue(fr, q ^ 1) : fr, q, inCount, seed, seed2, doIndirect | (r->fastShade ? 2 : 0), outCount, shCount);
            evEnd2(r, ev, sx);
            // NEE shadow rays of this wave: third stream, beside the next wave's closest-hit launch.  The shadow queue is
            // rewritten by the NEXT shade_wave, which therefore waits for this launch (evJoin2).  (`shadow_on_wave` 1 keeps them on
            // the wave stream: equal at full size and for the windows of 4 / 8 ranks, 8 % slower for those of 2 ranks.)
            const bool shadowOnWave = r->shadowOnWave != 0 || twoWave;
            hipStream_t ss = (overlap && !shadowOnWave) ? r->aux2 : sx;
            LmScene scs = scx;
            if (ss != sx) { scs.spill += (size_t)r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS); LM_HIP(hipEventRecord(r->evShade[depth], sx)); LM_HIP(hipStreamWaitEvent(ss, r->evShade[depth], 0)); }
            evBegin2(r, 1, ev, ss);
            Z(ss); K->trace_shadow(ss, gridAux, scs, fr, shCount, 0.01f, r->refillBelow);     // tmin of the intersection launch (:843)
            evEnd2(r, ev, ss);
            if (overlap) { LM_HIP(hipEventRecord(r->evJoin2, ss)); }
        }
        q ^= 1;
        seed = wangHash(seed);                                                               // :830
    }
    if (overlap) {
        LM_HIP(hipEventRecord(r->evJoin, sx)); LM_HIP(hipStreamWaitEvent(st, r->evJoin, 0));
        if (depthMax > 1) LM_HIP(hipStreamWaitEvent(st, r->evJoin2, 0));
        if (tailLaunched) LM_HIP(hipStreamWaitEvent(st, r->evTail, 0));
    }
    Z(st); K->merge(st, r->gridFor(fr.n, 8), fr, (blend ? 1 : 0) | (lazy ? 2 : 0), r->blendCounter, (int)depthMax);     // + ReSTIR::SwapBuffers per executed wave
    if (r->pinnedCounters[par]) {     // asynchronous counter read-back: feeds the next frames' schedule (above); before evMerge, which
        // releases this counter block to the frame after next
        LM_HIP(hipMemcpyAsync(r->pinnedCounters[par], fr.counters, LM_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        LM_HIP(hipEventRecord(r->evCnt[par], st));
        r->cntPending[par] = true;
    }
    if (overlap) LM_HIP(hipEventRecord(r->evMerge[par], st));
    evEnd(r, evAll);
    LM_HIP(hipGetLastError());
    r->lastDepth = depthMax;
    r->lastLightCount = r->lights.size();
    if (blend) ++r->blendCounter;                                                            // :1039-1042
    r->frameIndex = r->frameIndex + 1 == 2 ? 0 : r->frameIndex + 1;                          // :1045-1049
    r->lastGbuf = r->gbufIndex; r->gbufIndex = (r->gbufIndex + 1) % 3;
    memcpy(r->prevCamWorld, camWorld, sizeof camWorld);                                      // :1051
    ++r->frameCount;                                                                         // :1052
    r->framesTraced.fetch_add(1);
    return 0;
}

int syncAndCollect(R* r)
{
    if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    LM_HIP(hipStreamSynchronize(r->stream));
    if (!r->countersValid && r->fr.counters) {
        LM_HIP(hipMemcpy(r->hostCounters, r->fr.counters, sizeof r->hostCounters, hipMemcpyDeviceToHost));
        r->countersValid = true;
    }
    if (r->evUsed) {
        // accumulate over every frame enqueued since the last lumen_mi_enable_kernel_timing(1)
        for (size_t i = 0; i < r->evUsed; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r->evPool[i].a, r->evPool[i].b) == hipSuccess) { r->classMs[r->evPool[i].cls] += ms; r->classLaunches[r->evPool[i].cls]++; }
        }
        r->evUsed = 0;
        const float frames = (float)std::max<uint32_t>(1u, r->classLaunches[4]);
        r->frameStats["Wavefront Iteration"] = (uint64_t)((r->classMs[0] + r->classMs[2] + r->classMs[3] + r->classMs[5]) * 1000.f / frames);
        r->frameStats["Shadow Rays"] = (uint64_t)(r->classMs[1] * 1000.f / frames);
        r->frameStats["ReSTIR"] = (uint64_t)(r->classMs[3] * 1000.f / frames);
        r->frameStats["Total Frame Time"] = (uint64_t)(r->classMs[4] * 1000.f / frames);
    }
    return 0;
}

}  // namespace lmr
