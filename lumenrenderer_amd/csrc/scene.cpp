// scene.cpp — host side of the scene: emissive classification, flattening into the scene data table + world-space triangle soup,
// BVH build and upload, the double-buffered scene sets (edits / GPU refit), resource upload and the light list.
// Reference: PTScene.cpp:74-156, PTMeshInstance.cpp:123-178, LightDataBuffer.cpp:37-125, GPUDataBufferKernels.cu:9-186.
#include "renderer_state.h"
#include <chrono>

namespace lmr {

// ---- host texture fetch (light-list build only; same definition as the device fetch) -------------------------
void texel(const Texture& t, int x, int y, float out[4])
{
    const uint32_t p = t.px[(size_t)y * t.w + x];
    const uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u, a = p >> 24;
    if (t.srgb) { out[0] = g_srgbLut[r]; out[1] = g_srgbLut[g]; out[2] = g_srgbLut[b]; } else { out[0] = (float)r / 255.0f; out[1] = (float)g / 255.0f; out[2] = (float)b / 255.0f; }
    out[3] = (float)a / 255.0f;
}
int wrapi(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
void tex2D(const R* r, int id, float u, float v, float out[4])
{
    if (id < 0) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
    const Texture& t = r->textures[id];
    if (t.w == 1 && t.h == 1) { texel(t, 0, 0, out); return; }
    const bool cudaRule = r->texFilter == 0;               // the same rule as lm_shade.h lm_tex2D (decision D6)
    const float x = (cudaRule ? u - floorf(u) : u) * (float)t.w - 0.5f, y = (cudaRule ? v - floorf(v) : v) * (float)t.h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    float ax = x - fx0, ay = y - fy0;
    if (cudaRule) { ax = floorf(ax * 256.0f + 0.5f) * (1.0f / 256.0f); ay = floorf(ay * 256.0f + 0.5f) * (1.0f / 256.0f); }
    const int x0 = wrapi((int)fx0, (int)t.w), y0 = wrapi((int)fy0, (int)t.h);
    const int x1 = wrapi(x0 + 1, (int)t.w), y1 = wrapi(y0 + 1, (int)t.h);
    float t00[4], t10[4], t01[4], t11[4];
    texel(t, x0, y0, t00); texel(t, x1, y0, t10); texel(t, x0, y1, t01); texel(t, x1, y1, t11);
    for (int k = 0; k < 4; k++) {
        const float a = t00[k] + ax * (t10[k] - t00[k]);
        const float b = t01[k] + ax * (t11[k] - t01[k]);
        out[k] = a + ay * (b - a);
    }
}

// FindEmissives — reference GPUEmissiveLookup.cu:13-109, gate WaveFrontRenderer.cpp:1192-1210
void findEmissives(const R* r, Primitive& p)
{
    const Material& m = r->materials[p.material];
    p.emissive.assign(p.idx.size() / 3, 0);
    p.numLights = 0;
    if (m.emissiveColor[0] == 0.f && m.emissiveColor[1] == 0.f && m.emissiveColor[2] == 0.f) { p.containEmissive = false; return; }
    for (size_t b = 0; b + 2 < p.idx.size(); b += 3) {
        const Vertex48 &v0 = p.verts[p.idx[b]], &v1 = p.verts[p.idx[b + 1]], &v2 = p.verts[p.idx[b + 2]];
        constexpr float oneThird = 1.f / 3.f;
        const float uvx = (v0.uv[0] + v1.uv[0] + v2.uv[0]) * oneThird, uvy = (v0.uv[1] + v1.uv[1] + v2.uv[1]) * oneThird;
        float e[4] = {m.dev.emissive.x, m.dev.emissive.y, m.dev.emissive.z, m.dev.emissive.w};
        if (m.dev.tex[4] >= 0) { float t[4]; tex2D(r, m.dev.tex[4], uvx, uvy, t); for (int k = 0; k < 4; k++) e[k] = e[k] * t[k]; }
        if (e[0] > 0.0f || e[1] > 0.0f || e[2] > 0.0f) { p.emissive[b / 3] = 1; p.numLights++; }
    }
    p.containEmissive = p.numLights > 0;
}

// scene data table + world-space triangle soup + BVH — replaces PTScene/PTMeshInstance/OptixWrapper AS builds
// Instance state changed (matrices, emissive mode / radiance, override material) but not the set of instances: refresh the
// host copy of the scene data table; syncScene() carries it (and, if something moved, a BVH refit on the GPU, kernels.hip
// "BVH refit") to the device.
int refreshEntries(R* r)
{
    const Scene& sc = r->scenes[r->activeScene];
    for (size_t ii : sc.instances) {
        const Instance& mi = r->instances[ii];
        const std::vector<size_t>& prims = r->meshes[mi.mesh].prims;
        for (size_t k = 0; k < mi.entries.size() && k < prims.size(); k++) {
            LmEntry& e = r->entries[mi.entries[k]];
            memcpy(e.m, mi.M, sizeof e.m);
            e.material = (uint32_t)(mi.overrideMaterial >= 0 ? (size_t)mi.overrideMaterial : r->prims[prims[k]].material);
            e.mode = (uint32_t)mi.mode;
            e.emissive = make_float4(mi.radiance[0], mi.radiance[1], mi.radiance[2], mi.scale);
        }
    }
    ++r->entriesVer;
    if (r->transformsDirty) { ++r->geomVer; r->worldTris.clear(); }          // (host copy of the world triangles: ensureWorldTris)
    r->transformsDirty = false;
    r->entriesDirty = false;
    r->lightsDirty = true;
    return 0;
}

// Bring the device scene up to the host state.  If the set the previous frames read is stale, the other set is written on
// stream `su` and becomes current: it was last read by a frame at least two back, whose merge `su` has already waited for
// (traceFrameAsync), so nothing in flight reads what is overwritten here.  No host synchronisation except for the reuse of
// a staging buffer whose previous copy (two scene states ago) has not finished yet.
// boxes and Woop packets of a scene set from its topology and the instance table: triangles re-transformed, packets recomputed (bit-identical to the host builder:
// lm_tri.h), boxes propagated bottom-up level by level and quantised against the scene box (kernels.hip lm_k_refit_*), top-of-tree table rebuilt
static int refitSet(R* r, SceneSet& T, hipStream_t su)
{
    const LmKernelTable* K = r->K;
    LmScene sc = r->dscene;
    sc.nodes = T.nodes.p; sc.packets = T.packets.p; sc.quant = T.quant.p; sc.entries = T.entries.p; sc.numEntries = (uint32_t)r->entries.size(); sc.triId = T.triId.p; sc.triOrder = T.triOrder.p;
    const uint32_t nt = T.nTris;
    if (r->dTriBox.ensure(2 * (size_t)nt + 2) || r->dNodeBox.ensure(2 * std::max<size_t>(T.nodes.cap, 1))) return fail(LUMEN_MI_ERR_DEVICE, "refit buffer allocation failed");
    K->refit_tris(su, sc, nt, r->dTriBox.p, r->dRefitBounds.p);
    K->refit_quant(su, r->dRefitBounds.p, T.quant.p);
    for (size_t l = 0; l + 1 < T.levelStart.size(); l++) {
        const uint32_t a = T.levelStart[l], b = T.levelStart[l + 1];
        if (b > a) K->refit_level(su, sc, T.levelNodes.p + a, b - a, r->dTriBox.p, r->dNodeBox.p);
    }
    if (T.top.ensure(LM_TOP_NODES + 1)) return fail(LUMEN_MI_ERR_DEVICE, "top table allocation failed");
    K->build_top(su, T.nodes.p, T.top.p);                   // the boxes changed: so does their copy in the top-of-tree table
    LM_HIP(hipGetLastError());
    return 0;
}

int syncScene(R* r, hipStream_t su)
{
    if (r->sset[0].nodes.p == nullptr) return 0;                   // nothing built yet
    SceneSet& C = r->sset[r->sgen];
    if (C.entriesVer != r->entriesVer || C.geomVer != r->geomVer || C.lightsVer != r->lightsVer || C.topoVer != r->topoVer) {
        SceneSet& T = r->sset[r->sgen ^ 1];
        if (T.upPending) { LM_HIP(hipEventSynchronize(T.evUp)); T.upPending = false; }
        if (!T.evUp) LM_HIP(hipEventCreateWithFlags(&T.evUp, hipEventDisableTiming));
        bool copied = false;
        if (T.topoVer != r->topoVer) {                              // a new tree (instances added / removed): topology now, boxes by the refit below
            const LmBvh& b = r->bvh;
            const size_t nn = b.nodesW.size(), ns = b.order.size(), nl = b.levelNodes.size();
            if (T.hNodes.ensure(nn) || T.hTriId.ensure(ns) || T.hOrder.ensure(ns) || T.hLevelNodes.ensure(nl) || T.nodes.ensure(nn) || T.triId.ensure(ns) ||
                T.triOrder.ensure(ns) || T.levelNodes.ensure(nl) || T.packets.ensure(ns + 1) || T.quant.ensure(8))
                return fail(LUMEN_MI_ERR_DEVICE, "scene tree allocation failed");
            memcpy(T.hNodes.p, b.nodesW.data(), nn * sizeof(LmNodeW)); memcpy(T.hTriId.p, r->triId.data(), ns * sizeof(uint2));
            memcpy(T.hOrder.p, b.order.data(), ns * sizeof(uint32_t)); memcpy(T.hLevelNodes.p, b.levelNodes.data(), nl * sizeof(uint32_t));
            LM_HIP(hipMemcpyAsync(T.nodes.p, T.hNodes.p, nn * sizeof(LmNodeW), hipMemcpyHostToDevice, su));
            if (ns) { LM_HIP(hipMemcpyAsync(T.triId.p, T.hTriId.p, ns * sizeof(uint2), hipMemcpyHostToDevice, su)); LM_HIP(hipMemcpyAsync(T.triOrder.p, T.hOrder.p, ns * sizeof(uint32_t), hipMemcpyHostToDevice, su)); }
            if (nl) LM_HIP(hipMemcpyAsync(T.levelNodes.p, T.hLevelNodes.p, nl * sizeof(uint32_t), hipMemcpyHostToDevice, su));
            LM_HIP(hipMemsetAsync(T.packets.p + ns, 0, sizeof(LmTriPacket), su));            // sentinel packet
            T.levelStart = b.levelStart; T.nTris = (uint32_t)ns;
            T.topoVer = r->topoVer; T.geomVer = 0;                  // no boxes yet
            copied = true;
        }
        if (T.entriesVer != r->entriesVer) {
            const size_t n = r->entries.size();
            if (T.hEntries.ensure(n) || T.entries.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "scene table allocation failed");
            if (n) { memcpy(T.hEntries.p, r->entries.data(), n * sizeof(LmEntry)); LM_HIP(hipMemcpyAsync(T.entries.p, T.hEntries.p, n * sizeof(LmEntry), hipMemcpyHostToDevice, su)); copied = true; }
            T.entriesVer = r->entriesVer;
        }
        if (T.lightsVer != r->lightsVer) {
            const size_t n = r->lights.size();
            if (T.hLights.ensure(n) || T.hCdf.ensure(n) || T.lights.ensure(n) || T.cdf.ensure(n)) return fail(LUMEN_MI_ERR_DEVICE, "light list allocation failed");
            if (n) {
                memcpy(T.hLights.p, r->lights.data(), n * sizeof(LmLight)); memcpy(T.hCdf.p, r->cdf.data(), n * sizeof(float));
                LM_HIP(hipMemcpyAsync(T.lights.p, T.hLights.p, n * sizeof(LmLight), hipMemcpyHostToDevice, su));
                LM_HIP(hipMemcpyAsync(T.cdf.p, T.hCdf.p, n * sizeof(float), hipMemcpyHostToDevice, su));
                copied = true;
            }
            T.lightsVer = r->lightsVer;
        }
        if (copied) { LM_HIP(hipEventRecord(T.evUp, su)); T.upPending = true; }
        if (T.geomVer != r->geomVer) {
            int rc = refitSet(r, T, su); if (rc) return rc;
            ++r->refits;
            T.geomVer = r->geomVer;
        }
        r->sgen ^= 1;
    }
    const SceneSet& S = r->sset[r->sgen];
    r->dscene.nodes = S.nodes.p; r->dscene.packets = S.packets.p; r->dscene.quant = S.quant.p; r->dscene.entries = S.entries.p; r->dscene.numEntries = (uint32_t)r->entries.size();
    r->dscene.triId = S.triId.p; r->dscene.triOrder = S.triOrder.p; r->dscene.top = S.top.p;
    r->dscene.lights = S.lights.p; r->dscene.cdf = S.cdf.p;
    return 0;
}

int flatten(R* r)
{
    if (!r->sceneDirty) {
        if (!r->transformsDirty && !r->entriesDirty) return 0;
        if (r->refitEnabled && r->activeScene >= 0 && !r->entries.empty()) return refreshEntries(r);
        r->sceneDirty = true;
    }
    if (r->activeScene < 0) return fail(LUMEN_MI_ERR_STATE, "no scene set (lumen_mi_set_scene)");
    Scene& sc = r->scenes[r->activeScene];
    r->entries.clear(); r->entryPrim.clear(); r->worldTris.clear(); r->triEntry.clear(); r->triPrim.clear();
    // A topology edit after the first build (instances added / removed) does not need a new SAH tree: every mesh keeps its own tree
    // (object space, built once), the scene tree is those trees behind a small top tree over the instances (lm_assemble_bvh), and
    // the GPU refit computes boxes and Woop packets from the instance transforms.  The first build is the full SAH build.
    const bool newPrims = r->poolPrims != r->prims.size();          // primitives are only ever appended
    const bool assemble = r->assembleEnabled && r->refitEnabled && r->builtOnce;
    std::vector<LmInstanceRef> refs;
    // vertex / index pools: one slot range per primitive (primitives are only ever appended: the pools are rebuilt and uploaded
    // when new ones exist)
    std::vector<float4> verts; std::vector<uint32_t> indices;
    if (newPrims) {
        r->vertBase.assign(r->prims.size(), 0); r->idxBase.assign(r->prims.size(), 0);
        for (size_t p = 0; p < r->prims.size(); p++) {
            r->vertBase[p] = (uint32_t)(verts.size() / 3); r->idxBase[p] = (uint32_t)indices.size();
            for (const Vertex48& v : r->prims[p].verts) {
                verts.push_back(make_float4(v.pos[0], v.pos[1], v.pos[2], v.uv[0]));
                verts.push_back(make_float4(v.uv[1], v.normal[0], v.normal[1], v.normal[2]));
                verts.push_back(make_float4(v.tangent[0], v.tangent[1], v.tangent[2], v.tangent[3]));
            }
            indices.insert(indices.end(), r->prims[p].idx.begin(), r->prims[p].idx.end());
        }
    }
    const std::vector<uint32_t>& vertBase = r->vertBase; const std::vector<uint32_t>& idxBase = r->idxBase;
    for (size_t ii : sc.instances) {
        Instance& mi = r->instances[ii];
        mi.entries.clear();
        if (assemble) {
            Mesh& mesh = r->meshes[mi.mesh];
            if (!mesh.bvh) {                                   // the mesh's own tree, triangles in the order the loop below numbers them
                std::vector<float> local;
                for (size_t p : mesh.prims) { const Primitive& pr = r->prims[p]; for (size_t t = 0; t + 2 < pr.idx.size(); t += 3) for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) local.push_back(pr.verts[pr.idx[t + k]].pos[a]); }
                mesh.tris = (uint32_t)(local.size() / 9);
                mesh.bvh = std::make_shared<LmBvh>();
                lm_build_bvh(local.data(), mesh.tris, mesh.bvh.get());
                // the assembly only needs the 4-wide topology and the triangle order: drop the binary tree, packets and level lists
                std::vector<LmNode>().swap(mesh.bvh->nodes); std::vector<LmTriPacket>().swap(mesh.bvh->packets);
                std::vector<uint32_t>().swap(mesh.bvh->levelNodes); std::vector<uint32_t>().swap(mesh.bvh->levelStart);
                for (int a = 0; a < 3; a++) { mesh.lo[a] = INFINITY; mesh.hi[a] = -INFINITY; }
                for (size_t f = 0; f < local.size(); f++) { mesh.lo[f % 3] = std::min(mesh.lo[f % 3], local[f]); mesh.hi[f % 3] = std::max(mesh.hi[f % 3], local[f]); }
            }
            if (mesh.tris) {
                LmInstanceRef ref; ref.mesh = mesh.bvh.get(); ref.triBase = (uint32_t)r->triEntry.size();
                for (int a = 0; a < 3; a++) { ref.box[a] = INFINITY; ref.box[3 + a] = -INFINITY; }
                for (int c = 0; c < 8; c++) {
                    const float corner[3] = {(c & 1) ? mesh.hi[0] : mesh.lo[0], (c & 2) ? mesh.hi[1] : mesh.lo[1], (c & 4) ? mesh.hi[2] : mesh.lo[2]};
                    float w[3]; mulPoint(mi.M, corner, 1.f, w);
                    for (int a = 0; a < 3; a++) { ref.box[a] = std::min(ref.box[a], w[a]); ref.box[3 + a] = std::max(ref.box[3 + a], w[a]); }
                }
                refs.push_back(ref);
            }
        }
        for (size_t p : r->meshes[mi.mesh].prims) {
            LmEntry e;
            memcpy(e.m, mi.M, sizeof e.m);
            e.vertBase = vertBase[p]; e.idxBase = idxBase[p];
            e.material = (uint32_t)(mi.overrideMaterial >= 0 ? (size_t)mi.overrideMaterial : r->prims[p].material);
            e.mode = (uint32_t)mi.mode;
            e.emissive = make_float4(mi.radiance[0], mi.radiance[1], mi.radiance[2], mi.scale);
            const uint32_t entryIdx = (uint32_t)r->entries.size();
            mi.entries.push_back(entryIdx);
            r->entries.push_back(e);
            r->entryPrim.push_back(p);
            const Primitive& pr = r->prims[p];
            for (size_t t = 0; t + 2 < pr.idx.size(); t += 3) {
                for (int k = 0; k < 3 && !assemble && r->gpuBuild <= 0; k++) {        // (neither the assembled nor the device-built tree needs host-side world triangles: ensureWorldTris makes them on demand)
                    float w[3];
                    mulPoint(e.m, pr.verts[pr.idx[t + k]].pos, 1.f, w);
                    r->worldTris.push_back(w[0]); r->worldTris.push_back(w[1]); r->worldTris.push_back(w[2]);
                }
                r->triEntry.push_back(entryIdx); r->triPrim.push_back((uint32_t)(t / 3));
            }
        }
    }
    const uint32_t nt = (uint32_t)r->triEntry.size();
    const bool assembled = assemble && !refs.empty();
    // tuning key gpu_build: the tree of a full build is made on the device (bvh_gpu.hip: Morton sort + radix tree + collapse; topology only, the refit kernels fill in
    // boxes and packets below) from the scene tables, which therefore go up first; any reason it cannot apply falls back to the host SAH builder
    bool gpuBuilt = false;
    if (!assembled && r->gpuBuild > 0 && nt >= 2u) {
        hipStream_t st = r->stream;
        if (hipSetDevice(r->device) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "stream sync failed");
        if (newPrims) {
            if (r->dVerts.upload(verts, st) || r->dIndices.upload(indices, st)) return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
            r->poolPrims = r->prims.size(); r->dscene.verts = r->dVerts.p; r->dscene.indices = r->dIndices.p;
        }
        std::vector<uint2> triIn(nt);
        for (uint32_t t = 0; t < nt; t++) triIn[t] = make_uint2(r->triEntry[t], r->triPrim[t]);
        DevBuf<LmEntry> dE; DevBuf<uint2> dT;
        const auto t0 = std::chrono::steady_clock::now();
        if (!dE.upload(r->entries, st) && !dT.upload(triIn, st)) {
            const int rcb = lm_build_bvh_gpu(st, dE.p, r->dVerts.p, r->dIndices.p, dT.p, nt, &r->bvh);
            gpuBuilt = rcb == 0;
            if (getenv("LUMEN_MI_BUILD_TIMING")) fprintf(stderr, "[bvh] device build of %u triangles: rc %d, %.3f s (uploads of the instance table and triangle ids included), %zu wide nodes, stack %u\n",
                                                         nt, rcb, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), r->bvh.nodesW.size(), r->bvh.maxStack);
        }
        dE.release(); dT.release();
        if (gpuBuilt) ++r->gpuBuilds;
    }
    if (assembled) { lm_assemble_bvh(refs.data(), (uint32_t)refs.size(), &r->bvh); ++r->assemblies; }
    else if (!gpuBuilt) { if (assemble || r->worldTris.size() != (size_t)9 * nt) ensureWorldTris(r); lm_build_bvh(r->worldTris.data(), nt, &r->bvh); }
    if (r->bvh.maxStack > LM_STACK_DEPTH) return fail(LUMEN_MI_ERR_STATE, "BVH needs a deeper traversal stack than LM_STACK_DEPTH");
    r->triId.resize(nt);
    for (uint32_t s = 0; s < nt; s++) r->triId[s] = make_uint2(r->triEntry[r->bvh.order[s]], r->triPrim[r->bvh.order[s]]);
    ++r->entriesVer; ++r->geomVer; ++r->topoVer;
    if (newPrims && r->poolPrims != r->prims.size()) {      // (the device build above has already put them up)
        // new geometry: the vertex / index pools grow (existing ranges keep their bytes).  Behind the merge of the last frame by
        // stream order, and the host waits, because a grown buffer is a new allocation.
        hipStream_t st = r->stream;
        if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "stream sync failed");
        if (r->dVerts.upload(verts, st) || r->dIndices.upload(indices, st)) return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
        if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "scene upload sync failed");
        r->poolPrims = r->prims.size();
        r->dscene.verts = r->dVerts.p; r->dscene.indices = r->dIndices.p;
    }
    if (assembled) {
        // nothing touches the device here: syncScene() carries the new tree into the idle scene set on the wave stream and refits it
        // there, like any other scene edit — frames keep overlapping while instances come and go
        r->sceneDirty = false; r->transformsDirty = false; r->entriesDirty = false; r->lightsDirty = true;
        return 0;
    }
    hipStream_t st = r->stream;
    // (stream order puts these copies behind the merge of the last frame, which has joined every other stream; the host then
    // waits for them, so both scene sets are idle and identical afterwards)
    if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "stream sync failed");
    for (SceneSet& S : r->sset) if (S.upPending) { (void)hipEventSynchronize(S.evUp); S.upPending = false; }
    std::vector<float> quant = {r->bvh.qmin[0], r->bvh.qmin[1], r->bvh.qmin[2], r->bvh.qstep[0], r->bvh.qstep[1], r->bvh.qstep[2], r->bvh.pad, 0.f};
    {
        std::vector<uint32_t> bounds = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
        if (r->dRefitBounds.upload(bounds, st) || r->dTriBox.ensure(2 * (size_t)nt + 2) || r->dNodeBox.ensure(2 * r->bvh.nodesW.size()))
            return fail(LUMEN_MI_ERR_DEVICE, "refit buffer allocation failed");
    }
    for (SceneSet& S : r->sset) {
        if (S.nodes.upload(r->bvh.nodesW, st) || S.entries.upload(r->entries, st) || S.quant.upload(quant, st) ||
            S.triId.upload(r->triId, st) || S.triOrder.upload(r->bvh.order, st) || S.levelNodes.upload(r->bvh.levelNodes, st))
            return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
        S.levelStart = r->bvh.levelStart; S.nTris = nt;
        S.entriesVer = r->entriesVer; S.geomVer = r->geomVer; S.topoVer = r->topoVer;
        if (gpuBuilt) {
            // the device builder produced the topology only: packets and boxes by the refit kernels, as after an instance-level assembly
            if (S.packets.ensure((size_t)nt + 1) || hipMemsetAsync(S.packets.p + nt, 0, sizeof(LmTriPacket), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "packet allocation failed");
            int rcr = refitSet(r, S, st); if (rcr) return rcr;
        } else {
            if (S.packets.upload(r->bvh.packets, st)) return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
            if (S.top.ensure(LM_TOP_NODES + 1)) return fail(LUMEN_MI_ERR_DEVICE, "top table allocation failed");
            r->K->build_top(st, S.nodes.p, S.top.p);
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "scene upload sync failed");
    if (r->dSpill.ensure((size_t)4 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS)))      // one area per stream
        return fail(LUMEN_MI_ERR_DEVICE, "stack spill allocation failed");
    r->dscene.spill = r->dSpill.p;
    r->dscene.verts = r->dVerts.p; r->dscene.indices = r->dIndices.p;
    {
        const SceneSet& S = r->sset[r->sgen];
        r->dscene.nodes = S.nodes.p; r->dscene.packets = S.packets.p; r->dscene.quant = S.quant.p; r->dscene.entries = S.entries.p; r->dscene.numEntries = (uint32_t)r->entries.size();
        r->dscene.triId = S.triId.p; r->dscene.triOrder = S.triOrder.p; r->dscene.top = S.top.p;
    }
    r->sceneDirty = false;
    r->transformsDirty = false;
    r->entriesDirty = false;
    r->lightsDirty = true;
    r->builtOnce = true;
    return 0;
}

// world-space triangle soup on the host: input of the SAH build, and what lumen_mi_get_world_triangles returns; an assembled
// scene tree does not need it, so it is produced on demand
void ensureWorldTris(R* r)
{
    if (r->worldTris.size() == 9 * r->triEntry.size()) return;
    r->worldTris.clear(); r->worldTris.reserve(9 * r->triEntry.size());
    for (size_t g = 0; g < r->triEntry.size(); g++) {
        const LmEntry& e = r->entries[r->triEntry[g]];
        const Primitive& pr = r->prims[r->entryPrim[r->triEntry[g]]];
        for (int k = 0; k < 3; k++) {
            float w[3];
            mulPoint(e.m, pr.verts[pr.idx[3 * (size_t)r->triPrim[g] + k]].pos, 1.f, w);
            r->worldTris.push_back(w[0]); r->worldTris.push_back(w[1]); r->worldTris.push_back(w[2]);
        }
    }
}

int uploadResources(R* r)
{
    hipStream_t st = r->stream;
    if (r->texturesDirty) {
        std::vector<LmTexDesc> desc; std::vector<uint32_t> texels;
        for (const Texture& t : r->textures) { desc.push_back(LmTexDesc{(uint32_t)texels.size(), t.w, t.h, (t.srgb ? 1u : 0u) | (r->texFilter ? 2u : 0u)}); texels.insert(texels.end(), t.px.begin(), t.px.end()); }
        std::vector<float> lut(g_srgbLut, g_srgbLut + 256);
        if (r->dTexDesc.upload(desc, st) || r->dTexels.upload(texels, st) || r->dLut.upload(lut, st)) return fail(LUMEN_MI_ERR_DEVICE, "texture upload failed");
        if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "texture upload sync failed");
        r->dscene.texDesc = r->dTexDesc.p; r->dscene.texels = r->dTexels.p; r->dscene.srgbLut = r->dLut.p;
        r->texturesDirty = false;
    }
    if (r->materialsDirty) {
        std::vector<LmDevMaterial> m;
        for (const Material& x : r->materials) m.push_back(x.dev);
        if (r->dMaterials.upload(m, st)) return fail(LUMEN_MI_ERR_DEVICE, "material upload failed");
        if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "material upload sync failed");
        r->dscene.materials = r->dMaterials.p; r->dscene.numMaterials = (uint32_t)r->materials.size();
        r->materialsDirty = false;
    }
    return 0;
}

// light list + CDF — reference LightDataBuffer.cpp:37-125, GPUDataBufferKernels.cu:9-186 (launch shape
// CPUDataBufferKernels.cu:35-56), ReSTIRKernels.cu:49-130 (sort by mean radiance, weights, inclusive scan).
// Built on the host and cached while the scene is unchanged (the reference rebuilds both every frame).
int buildLights(R* r)
{
    if (!r->lightsDirty) return 0;
    struct LID { uint32_t tableIndex, numTriangles, numEmissives; };
    std::vector<LID> lid;
    uint32_t numEmissivePrims = 0, total = 0;
    float avg = 0;
    const Scene& sc = r->scenes[r->activeScene];
    for (size_t ii : sc.instances) {
        const Instance& mi = r->instances[ii];
        bool meshEmissive = false;
        for (size_t p : r->meshes[mi.mesh].prims) meshEmissive |= r->prims[p].containEmissive;
        if (mi.mode != 1 && ((mi.mode == 0 && meshEmissive) || mi.mode == 2)) {
            for (size_t k = 0; k < r->meshes[mi.mesh].prims.size(); k++) {
                const Primitive& pr = r->prims[r->meshes[mi.mesh].prims[k]];
                if (pr.containEmissive || mi.mode == 2) {
                    const uint32_t numTriangles = (uint32_t)(pr.idx.size() / 3);
                    avg = ((avg * (float)numEmissivePrims) + (float)numTriangles) / (float)(numEmissivePrims + 1);
                    numEmissivePrims++;
                    total += pr.numLights;
                    lid.push_back(LID{mi.entries[k], numTriangles, pr.numLights});
                }
            }
        }
    }
    const uint32_t bufferSize = 1000000u;                       // LightDataBuffer(1'000'000), WaveFrontRenderer.cpp:295
    if (total > bufferSize) {
        size_t keep = lid.size();
        while (keep > 0) { total -= lid[keep - 1].numEmissives; keep--; if (total < bufferSize) break; }
        lid.resize(keep);
    }
    const uint32_t avgTri = (uint32_t)roundf(avg);
    const uint32_t gridH = (uint32_t)ceilf((float)avgTri / 64.f);
    const uint32_t threadsY = gridH * 64u;
    std::vector<LmLight> L;
    for (const LID& d : lid) {
        if (threadsY == 0) break;
        const uint32_t perThread = (uint32_t)ceilf((float)d.numTriangles / (float)threadsY);
        const LmEntry& e = r->entries[d.tableIndex];
        const Primitive& pr = r->prims[r->entryPrim[d.tableIndex]];
        const Material& mat = r->materials[e.material];
        for (uint32_t ty = 0; ty < threadsY; ty++) {
            const uint32_t start = ty * perThread;
            if (!(start < d.numTriangles - 1u)) continue;      // reference behaviour: a slice that starts at the last triangle is dropped (GPUDataBufferKernels.cu:37)
            const uint32_t num = (start + perThread) < d.numTriangles ? perThread : d.numTriangles - start;
            for (uint32_t k = 0; k < num; k++) {
                const uint32_t tri = start + k;
                LmLight out; memset(&out, 0, sizeof out);       // reserved slot that is never set: zero light
                if ((e.mode == 0u && pr.emissive[tri]) || e.mode == 2u) {
                    const Vertex48 &v0 = pr.verts[pr.idx[tri * 3]], &v1 = pr.verts[pr.idx[tri * 3 + 1]], &v2 = pr.verts[pr.idx[tri * 3 + 2]];
                    float p0[3], p1[3], p2[3];
                    mulPoint(e.m, v0.pos, 1.f, p0); mulPoint(e.m, v1.pos, 1.f, p1); mulPoint(e.m, v2.pos, 1.f, p2);
                    constexpr float oneThird = 1.f / 3.f;
                    const float uvx = (v0.uv[0] + v1.uv[0] + v2.uv[0]) * oneThird, uvy = (v0.uv[1] + v1.uv[1] + v2.uv[1]) * oneThird;
                    float em[4] = {0, 0, 0, 0};
                    if (e.mode == 0u) {
                        float t[4]; tex2D(r, mat.dev.tex[4], uvx, uvy, t);
                        const float me[4] = {mat.dev.emissive.x * e.emissive.w, mat.dev.emissive.y * e.emissive.w, mat.dev.emissive.z * e.emissive.w, mat.dev.emissive.w * e.emissive.w};
                        for (int q = 0; q < 4; q++) em[q] = t[q] * me[q];
                    } else {
                        em[0] = e.emissive.x * e.emissive.w; em[1] = e.emissive.y * e.emissive.w; em[2] = e.emissive.z * e.emissive.w; em[3] = e.emissive.w * e.emissive.w;
                    }
                    if (em[0] > 0.f || em[1] > 0.f || em[2] > 0.f) {
                        const float nl[3] = {(v0.normal[0] + v1.normal[0] + v2.normal[0]) * oneThird, (v0.normal[1] + v1.normal[1] + v2.normal[1]) * oneThird,
                                             (v0.normal[2] + v1.normal[2] + v2.normal[2]) * oneThird};
                        float nw[3];
                        mulPoint(e.m, nl, 0.f, nw);
                        const float inv = 1.0f / sqrtf(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
                        nw[0] *= inv; nw[1] *= inv; nw[2] *= inv;
                        const float a[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]}, b[3] = {p0[0] - p2[0], p0[1] - p2[1], p0[2] - p2[2]};
                        const float cx = (a[1] * b[2] - b[1] * a[2]), cy = (a[0] * b[2] - b[0] * a[2]), cz = (a[0] * b[1] - b[0] * a[1]);
                        const float area = sqrtf(cx * cx + cy * cy + cz * cz) / 2.0f;
                        out.a = make_float4(p0[0], p0[1], p0[2], p1[0]);
                        out.b = make_float4(p1[1], p1[2], p2[0], p2[1]);
                        out.c = make_float4(p2[2], nw[0], nw[1], nw[2]);
                        out.d = make_float4(em[0], em[1], em[2], area);
                    }
                }
                L.push_back(out);
            }
        }
    }
    auto key = [](const LmLight& l) { return (l.d.x + l.d.y + l.d.z) / 3.f; };
    std::stable_sort(L.begin(), L.end(), [&](const LmLight& a, const LmLight& b) { return key(a) < key(b); });
    r->cdf.resize(L.size());
    double acc = 0;
    for (size_t i = 0; i < L.size(); i++) { acc += (double)key(L[i]); r->cdf[i] = (float)acc; }
    r->lights.swap(L);
    r->totalEmissive = total;
    ++r->lightsVer;                                             // syncScene() uploads the list
    r->dscene.numLights = (uint32_t)r->lights.size();
    r->dscene.cdfSum = r->cdf.empty() ? 0.f : r->cdf.back();
    r->lightsDirty = false;
    return 0;
}

}  // namespace lmr
