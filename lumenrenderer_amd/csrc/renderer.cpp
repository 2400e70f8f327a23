// renderer.cpp — host side of liblumen_mi.so: resource tables, scene flattening, light list, frame loop, C ABI.
//
// Mirrors the call surface of the reference's WaveFront::WaveFrontRenderer : LumenRenderer
// (LumenPT/src/Framework/WaveFrontRenderer.{h,cpp}); every extern "C" entry point is declared and cited in
// include/lumen_mi.h.  The frame loop follows WaveFrontRenderer::TraceFrame (.cpp:435-1089) for order of
// operations, seed evolution and counters, but enqueues the whole frame on one HIP stream without host round
// trips (the reference synchronises ~40 times per frame).
#include "../../include/lumen_mi.h"
#include "bvh.h"
#include "lm_launch.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void lm_read_pushes(hipStream_t s, unsigned long long* out);

namespace {

thread_local std::string g_lastError;
int fail(int code, const std::string& msg) { g_lastError = msg; return code; }

#define LM_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---- handles: type tag in the top byte ---------------------------------------------------------------------
enum HType : uint64_t { H_TEXTURE = 1, H_MATERIAL = 2, H_PRIMITIVE = 3, H_MESH = 4, H_SCENE = 5, H_INSTANCE = 6 };
inline lumen_mi_handle mkh(HType t, size_t idx) { return ((uint64_t)t << 56) | (uint64_t)(idx + 1); }
inline bool unh(lumen_mi_handle h, HType t, size_t n, size_t& idx) { if ((h >> 56) != (uint64_t)t) return false; idx = (size_t)(h & 0x00ffffffffffffffull); if (idx == 0 || idx > n) return false; idx--; return true; }

struct Vertex48 { float pos[3]; float uv[2]; float normal[3]; float tangent[4]; };
static_assert(sizeof(Vertex48) == 48, "Vertex layout (ModelStructs.h:21-28)");

struct Texture { uint32_t w, h; bool srgb; std::vector<uint32_t> px; };
struct Material { LmDevMaterial dev; float emissiveColor[3]; };
struct Primitive { std::vector<Vertex48> verts; std::vector<uint32_t> idx; size_t material; std::vector<uint8_t> emissive; uint32_t numLights = 0; bool containEmissive = false; };
struct Mesh { std::vector<size_t> prims; };
struct Instance { size_t scene; size_t mesh; float M[16]; int mode; float radiance[3]; float scale; long overrideMaterial; std::vector<uint32_t> entries; };
struct Scene { std::vector<size_t> instances; };

template <class T> struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return 1;
        cap = n; return 0;
    }
    int upload(const std::vector<T>& v, hipStream_t s) {
        if (ensure(v.size())) return 1;
        if (!v.empty() && hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return 1;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class T> struct HostBuf {                 // pinned staging memory for asynchronous uploads
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) return 1;
        cap = n; return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// What a moving / edited scene rewrites every frame exists twice.  A frame reads one set; the next scene state is written into
// the other one (instance table and light list by asynchronous copies from pinned memory, BVH boxes and Woop packets by the
// refit kernels) on the wave stream, so frames keep overlapping while the scene changes.
struct SceneSet {
    DevBuf<LmNode4> nodes; DevBuf<LmWoop> woop; DevBuf<float> quant; DevBuf<LmEntry> entries; DevBuf<LmLight> lights; DevBuf<float> cdf;
    HostBuf<LmEntry> hEntries; HostBuf<LmLight> hLights; HostBuf<float> hCdf;
    hipEvent_t evUp = nullptr; bool upPending = false;      // the staging buffers are free again once this event has passed
    uint64_t entriesVer = 0, geomVer = 0, lightsVer = 0;    // state of the host scene this set holds
    void release() {
        nodes.release(); woop.release(); quant.release(); entries.release(); lights.release(); cdf.release();
        hEntries.release(); hLights.release(); hCdf.release();
        if (evUp) { (void)hipEventDestroy(evUp); evUp = nullptr; }
    }
};

float g_srgbLut[256];
void initLut() { static bool d = false; if (d) return; for (int i = 0; i < 256; i++) { const double c = i / 255.0; g_srgbLut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4)); } d = true; }

inline uint32_t wangHash(uint32_t s) { s = (s ^ 61u) ^ (s >> 16); s *= 9u; s = s ^ (s >> 4); s *= 0x27d4eb2du; s = s ^ (s >> 15); return s; }
inline void pack8(uint32_t& w, uint32_t shift, float v) { const uint32_t q = (uint32_t)(v * 255.f); w &= ~(255u << shift); w |= q << shift; }

inline void mulPoint(const float* m, const float* v, float w, float* out)     // rows 0..2, operation order of sutil Matrix4x4 * float4
{
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3] * w;
    out[1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7] * w;
    out[2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11] * w;
}

}  // namespace

struct lumen_mi_renderer {
    bool initialised = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;              // second stream: the indirect waves run beside ReSTIR (both depend only on the depth-0 G-buffer)
    hipStream_t aux3 = nullptr;             // fourth stream: second ReSTIR visibility pass beside the second spatial pass
    hipStream_t aux2 = nullptr;             // third stream: NEE shadow rays of wave d run beside the closest-hit launch of wave d+1
    hipEvent_t evJoin = nullptr, evJoin2 = nullptr, evVis = nullptr, evVisDone = nullptr;
    hipEvent_t evPick = nullptr;
    int pickAhead = 1;                      // 1 on (default), 0 off, -1 only for windows under 1 Mpixel
    int shadowOnWave = 0;                   // 1: NEE shadow rays on the wave stream (the path tail then has the third stream to itself); measured: 8 % slower for half-frame windows, equal elsewhere
    hipEvent_t evFront = nullptr, evTemporal[2] = {nullptr, nullptr}, evTop = nullptr, evMerge[2] = {nullptr, nullptr};   // cross-frame pipelining (traceFrameAsync)
    int framePar = 0;                       // parity of the frame being enqueued: selects the channel buffers and the counter block
    bool fenceNeeded = true;                // main-stream work (uploads, memsets) the frame front on the aux stream must wait for
    std::vector<hipEvent_t> evShade;        // per wave: shade_wave(d) done
    int auxPriority = 1;                    // 1: highest priority for the aux streams, 0: default
    int aux3Priority = 0;                   // the visibility / pick-ahead stream runs at default priority (pick-ahead must not starve the main chain)
    bool overlap = true;
    int traceBlocksMain = 8, traceBlocksAux = 8;
    int numCU = 256;
    const LmKernelTable* K = nullptr;
    bool instrumented = false;
    int tailBelow = -1;                     // waves expected to hold fewer rays than this run as one path-tail launch (0 = off,
                                            // -1 = auto: 65536 for windows under 1 Mpixel, where the wave chain is the critical path, else 16384) ...
    int tailLanes = 16;                     // ... with this many paths per wavefront
    uint32_t* pinnedCounters[2] = {nullptr, nullptr}; hipEvent_t evCnt[2] = {nullptr, nullptr}; bool cntPending[2] = {false, false};
    uint32_t estRays[LM_MAX_DEPTH + 1] = {0}; bool haveEst = false;     // rays per wave of the most recent frame that has been read back
    int refillBelow = 40, refillVisibility = 32, refillPrimary = 0;      // lane-refill thresholds of the queue traversal kernels (tunable via LUMEN_MI_REFILL*)

    lumen_mi_settings settings{};
    lumen_mi_settings pending{};
    std::mutex settingsMutex;
    std::recursive_mutex frameMutex;        // every entry point that touches scene / frame state takes it (factories call each other)
    std::atomic<int> waiters{0};            // callers queued on frameMutex: the render thread lets them in between two frames

    std::vector<Texture> textures;
    std::vector<Material> materials;
    std::vector<Primitive> prims;
    std::vector<Mesh> meshes;
    std::vector<Scene> scenes;
    std::vector<Instance> instances;
    long activeScene = -1;
    bool sceneDirty = true, texturesDirty = true, materialsDirty = true;
    bool transformsDirty = false;           // only instance matrices changed since the last build: the BVH is refitted on the GPU
    bool entriesDirty = false;              // emissive mode / radiance / override material of an instance changed: scene table + lights only
    uint32_t refits = 0;                    // refits since the last full build
    int refitEnabled = 1;                   // 0: every transform change triggers a full host rebuild

    // camera
    float camPos[3] = {0, 0, 0}, camRight[3] = {-1, 0, 0}, camUp[3] = {0, 1, 0}, camFwd[3] = {0, 0, 1};
    float fovY = 90.f;
    float prevCamWorld[16]; bool havePrev = false;

    // window
    uint32_t wx0 = 0, wy0 = 0, wx1 = 0, wy1 = 0; bool windowSet = false;
    uint32_t ox0 = 0, oy0 = 0, ox1 = 0, oy1 = 0; bool tileSet = false;      // owned tile inside the window (global pixel coordinates)

    // persistent state (WaveFrontRenderer members)
    uint32_t frameCount = 0, blendCounter = 0;
    int frameIndex = 0;
    int gbufIndex = 0, lastGbuf = 0;        // physical G-buffer set of the frame being enqueued / of the last enqueued frame (3 sets)
    DevBuf<int> dSwap;                      // ReSTIR swap-chain index lives on the device (LmFrame::swap)

    // flattened scene (host)
    std::vector<LmEntry> entries;
    std::vector<size_t> entryPrim;          // table entry -> primitive
    std::vector<float> worldTris;
    std::vector<uint32_t> triEntry, triPrim;
    LmBvh bvh;
    std::vector<LmLight> lights; std::vector<float> cdf;
    uint32_t totalEmissive = 0;
    bool lightsDirty = true;

    // device scene
    SceneSet sset[2];                       // sset[sgen] is what the next frame's kernels read
    int sgen = 0;
    uint64_t entriesVer = 1, geomVer = 1, lightsVer = 1;      // versions of the host-side scene state (instance table, geometry, light list)
    DevBuf<uint2> dTriId; DevBuf<uint32_t> dTriOrder;
    DevBuf<float4> dVerts; DevBuf<uint32_t> dIndices; DevBuf<LmDevMaterial> dMaterials;
    DevBuf<float4> dTriBox, dNodeBox; DevBuf<uint32_t> dLevelNodes, dRefitBounds;
    DevBuf<int> dSpill; DevBuf<LmTexDesc> dTexDesc; DevBuf<uint32_t> dTexels; DevBuf<float> dLut;
    LmScene dscene{};

    // device frame
    LmFrame fr{};
    uint32_t allocN = 0, allocDepth = 0;
    DevBuf<float4> dTailRay[6];             // ray queue of the path tail, double-buffered by frame parity (3 planes each)
    hipEvent_t evTail = nullptr;
    DevBuf<float4> dRay[6], dSh[3], dSh2[4], dGbuf[3], dProbe[3], dRes[5], dResC[5], dDirect[2], dIndirect[2], dCombined;
    DevBuf<uint4> dHits; DevBuf<uint32_t> dMotion[2], dCounters; DevBuf<uchar4> dOutput; DevBuf<uint2> dBags;
    uint32_t hostCounters[LM_CNT_WORDS] = {0};
    bool countersValid = false;
    uint32_t lastDepth = 0;
    size_t lastLightCount = 0;

    // timing
    bool timing = false;
    struct EvPair { hipEvent_t a, b; int cls; };
    std::vector<EvPair> evPool; size_t evUsed = 0;
    float classMs[5] = {0}; uint32_t classLaunches[5] = {0};
    std::map<std::string, uint64_t> frameStats;

    // render thread
    std::thread renderThread; std::atomic<bool> stopFlag{false};

    int traceGrid() const { return numCU * 8; }
    int gridFor(uint32_t n, int perCU) const { const int full = (int)((n + 255u) / 256u); return std::max(1, std::min(full, numCU * perCU)); }
};

// frame mutex with a waiter count: std::mutex is not fair, and the render thread re-acquires it back to back
struct ApiLock {
    lumen_mi_renderer* r;
    explicit ApiLock(lumen_mi_renderer* r_) : r(r_) { r->waiters.fetch_add(1); r->frameMutex.lock(); r->waiters.fetch_sub(1); }
    ~ApiLock() { r->frameMutex.unlock(); }
    ApiLock(const ApiLock&) = delete; ApiLock& operator=(const ApiLock&) = delete;
};

namespace {

using R = lumen_mi_renderer;

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
    const float x = u * (float)t.w - 0.5f, y = v * (float)t.h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, ay = y - fy0;
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
    if (r->transformsDirty) ++r->geomVer;
    r->transformsDirty = false;
    r->entriesDirty = false;
    r->lightsDirty = true;
    return 0;
}

// Bring the device scene up to the host state.  If the set the previous frames read is stale, the other set is written on
// stream `su` and becomes current: it was last read by a frame at least two back, whose merge `su` has already waited for
// (traceFrameAsync), so nothing in flight reads what is overwritten here.  No host synchronisation except for the reuse of
// a staging buffer whose previous copy (two scene states ago) has not finished yet.
int syncScene(R* r, hipStream_t su)
{
    if (r->sset[0].nodes.p == nullptr) return 0;                   // nothing built yet
    SceneSet& C = r->sset[r->sgen];
    if (C.entriesVer != r->entriesVer || C.geomVer != r->geomVer || C.lightsVer != r->lightsVer) {
        SceneSet& T = r->sset[r->sgen ^ 1];
        if (T.upPending) { LM_HIP(hipEventSynchronize(T.evUp)); T.upPending = false; }
        if (!T.evUp) LM_HIP(hipEventCreateWithFlags(&T.evUp, hipEventDisableTiming));
        bool copied = false;
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
            const LmKernelTable* K = r->K;
            LmScene sc = r->dscene;
            sc.nodes = T.nodes.p; sc.woop = T.woop.p; sc.quant = T.quant.p; sc.entries = T.entries.p;
            const uint32_t nt = (uint32_t)r->bvh.order.size();
            K->refit_tris(su, sc, nt, r->dTriBox.p, r->dRefitBounds.p);
            K->refit_quant(su, r->dRefitBounds.p, T.quant.p);
            for (size_t l = 0; l + 1 < r->bvh.levelStart.size(); l++) {
                const uint32_t a = r->bvh.levelStart[l], b = r->bvh.levelStart[l + 1];
                if (b > a) K->refit_level(su, sc, r->dLevelNodes.p + a, b - a, r->dTriBox.p, r->dNodeBox.p);
            }
            LM_HIP(hipGetLastError());
            ++r->refits;
            T.geomVer = r->geomVer;
        }
        r->sgen ^= 1;
    }
    const SceneSet& S = r->sset[r->sgen];
    r->dscene.nodes = S.nodes.p; r->dscene.woop = S.woop.p; r->dscene.quant = S.quant.p; r->dscene.entries = S.entries.p;
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
    // vertex / index pools: one slot range per primitive
    std::vector<uint32_t> vertBase(r->prims.size()), idxBase(r->prims.size());
    std::vector<float4> verts; std::vector<uint32_t> indices;
    for (size_t p = 0; p < r->prims.size(); p++) {
        vertBase[p] = (uint32_t)(verts.size() / 3); idxBase[p] = (uint32_t)indices.size();
        for (const Vertex48& v : r->prims[p].verts) {
            verts.push_back(make_float4(v.pos[0], v.pos[1], v.pos[2], v.uv[0]));
            verts.push_back(make_float4(v.uv[1], v.normal[0], v.normal[1], v.normal[2]));
            verts.push_back(make_float4(v.tangent[0], v.tangent[1], v.tangent[2], v.tangent[3]));
        }
        indices.insert(indices.end(), r->prims[p].idx.begin(), r->prims[p].idx.end());
    }
    for (size_t ii : sc.instances) {
        Instance& mi = r->instances[ii];
        mi.entries.clear();
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
                for (int k = 0; k < 3; k++) {
                    float w[3];
                    mulPoint(e.m, pr.verts[pr.idx[t + k]].pos, 1.f, w);
                    r->worldTris.push_back(w[0]); r->worldTris.push_back(w[1]); r->worldTris.push_back(w[2]);
                }
                r->triEntry.push_back(entryIdx); r->triPrim.push_back((uint32_t)(t / 3));
            }
        }
    }
    const uint32_t nt = (uint32_t)r->triEntry.size();
    lm_build_bvh(r->worldTris.data(), nt, &r->bvh);
    if (r->bvh.maxStack > LM_STACK_DEPTH) return fail(LUMEN_MI_ERR_STATE, "BVH needs a deeper traversal stack than LM_STACK_DEPTH");
    std::vector<uint2> triId(nt);
    for (uint32_t s = 0; s < nt; s++) triId[s] = make_uint2(r->triEntry[r->bvh.order[s]], r->triPrim[r->bvh.order[s]]);
    hipStream_t st = r->stream;
    // (stream order puts these copies behind the merge of the last frame, which has joined every other stream; the host then
    // waits for them, so both scene sets are idle and identical afterwards)
    if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "stream sync failed");
    for (SceneSet& S : r->sset) if (S.upPending) { (void)hipEventSynchronize(S.evUp); S.upPending = false; }
    std::vector<float> quant = {r->bvh.qmin[0], r->bvh.qmin[1], r->bvh.qmin[2], r->bvh.qstep[0], r->bvh.qstep[1], r->bvh.qstep[2], r->bvh.pad, 0.f};
    ++r->entriesVer; ++r->geomVer;
    for (SceneSet& S : r->sset) {
        if (S.nodes.upload(r->bvh.nodes4, st) || S.woop.upload(r->bvh.woop, st) || S.entries.upload(r->entries, st) || S.quant.upload(quant, st))
            return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
        S.entriesVer = r->entriesVer; S.geomVer = r->geomVer;
    }
    if (r->dTriId.upload(triId, st) || r->dTriOrder.upload(r->bvh.order, st) || r->dVerts.upload(verts, st) || r->dIndices.upload(indices, st))
        return fail(LUMEN_MI_ERR_DEVICE, "scene upload failed (hipMalloc/hipMemcpy)");
    {
        std::vector<uint32_t> bounds = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
        if (r->dRefitBounds.upload(bounds, st) || r->dLevelNodes.upload(r->bvh.levelNodes, st) ||
            r->dTriBox.ensure(2 * (size_t)nt + 2) || r->dNodeBox.ensure(2 * r->bvh.nodes4.size()))
            return fail(LUMEN_MI_ERR_DEVICE, "refit buffer allocation failed");
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "scene upload sync failed");
    if (r->dSpill.ensure((size_t)4 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS)))      // one area per stream
        return fail(LUMEN_MI_ERR_DEVICE, "stack spill allocation failed");
    r->dscene.spill = r->dSpill.p;
    r->dscene.triId = r->dTriId.p; r->dscene.triOrder = r->dTriOrder.p;
    r->dscene.verts = r->dVerts.p; r->dscene.indices = r->dIndices.p;
    {
        const SceneSet& S = r->sset[r->sgen];
        r->dscene.nodes = S.nodes.p; r->dscene.woop = S.woop.p; r->dscene.quant = S.quant.p; r->dscene.entries = S.entries.p;
    }
    r->sceneDirty = false;
    r->transformsDirty = false;
    r->entriesDirty = false;
    r->lightsDirty = true;
    return 0;
}

int uploadResources(R* r)
{
    hipStream_t st = r->stream;
    if (r->texturesDirty) {
        std::vector<LmTexDesc> desc; std::vector<uint32_t> texels;
        for (const Texture& t : r->textures) { desc.push_back(LmTexDesc{(uint32_t)texels.size(), t.w, t.h, t.srgb ? 1u : 0u}); texels.insert(texels.end(), t.px.begin(), t.px.end()); }
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
        r->dscene.materials = r->dMaterials.p;
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

int ensureFrameBuffers(R* r)
{
    const uint32_t W = r->settings.render_width, H = r->settings.render_height;
    if (!r->windowSet) { r->wx0 = 0; r->wy0 = 0; r->wx1 = W; r->wy1 = H; }
    if (r->wx1 > W || r->wy1 > H || r->wx0 >= r->wx1 || r->wy0 >= r->wy1) return fail(LUMEN_MI_ERR_INVALID, "render window outside the image");
    const uint32_t ww = r->wx1 - r->wx0, wh = r->wy1 - r->wy0, n = ww * wh;
    LmFrame& f = r->fr;
    const bool realloc = n != r->allocN || f.W != W || f.H != H || f.x0 != r->wx0 || f.y0 != r->wy0 || f.ww != ww;
    f.W = W; f.H = H; f.x0 = r->wx0; f.y0 = r->wy0; f.ww = ww; f.wh = wh; f.n = n;
    if (r->tileSet && (r->ox0 < r->wx0 || r->oy0 < r->wy0 || r->ox1 > r->wx1 || r->oy1 > r->wy1)) return fail(LUMEN_MI_ERR_INVALID, "owned tile outside the render window");
    f.tx0 = r->tileSet ? r->ox0 - r->wx0 : 0; f.ty0 = r->tileSet ? r->oy0 - r->wy0 : 0; f.tx1 = r->tileSet ? r->ox1 - r->wx0 : ww; f.ty1 = r->tileSet ? r->oy1 - r->wy0 : wh;
    if (!realloc) return 0;
    int bad = 0;
    for (int i = 0; i < 6; i++) bad |= r->dRay[i].ensure(n) | r->dTailRay[i].ensure(n);
    for (int i = 0; i < 3; i++) bad |= r->dSh[i].ensure(n);
    for (int i = 0; i < 4; i++) bad |= r->dSh2[i].ensure(n);
    for (int i = 0; i < 3; i++) bad |= r->dGbuf[i].ensure((size_t)8 * n) | r->dProbe[i].ensure(n);
    for (int i = 0; i < 2; i++) bad |= r->dMotion[i].ensure(n);
    for (int i = 0; i < 5; i++) bad |= r->dRes[i].ensure((size_t)4 * n) | r->dResC[i].ensure(n);
    for (int i = 0; i < 2; i++) bad |= r->dDirect[i].ensure(n) | r->dIndirect[i].ensure(n);
    bad |= r->dCombined.ensure(n) | r->dHits.ensure(n) | r->dOutput.ensure(n);
    bad |= r->dCounters.ensure(2 * LM_CNT_WORDS) | r->dBags.ensure(50 * 1000);
    if (bad) return fail(LUMEN_MI_ERR_DEVICE, "frame buffer allocation failed");
    for (int q = 0; q < 2; q++) { f.rayO[q] = r->dRay[3 * q].p; f.rayD[q] = r->dRay[3 * q + 1].p; f.rayC[q] = r->dRay[3 * q + 2].p; }
    f.shO = r->dSh[0].p; f.shD = r->dSh[1].p; f.shR = r->dSh[2].p;
    f.visO = r->dSh2[0].p; f.visD = r->dSh2[1].p; f.vis2O = r->dSh2[2].p; f.vis2D = r->dSh2[3].p;
    f.hits = r->dHits.p;
    for (int i = 0; i < 3; i++) { f.gbuf[i] = r->dGbuf[i].p; f.probe[i] = r->dProbe[i].p; }
    for (int i = 0; i < 5; i++) { f.res[i] = r->dRes[i].p; f.resC[i] = r->dResC[i].p; }
    f.motion = r->dMotion[0].p; f.direct = r->dDirect[0].p; f.indirect = r->dIndirect[0].p; f.combined = r->dCombined.p; f.output = r->dOutput.p;
    f.counters = r->dCounters.p; f.bags = r->dBags.p;
    // ResizeBuffers (WaveFrontRenderer.cpp:1424-1540): history is dropped; reservoirs reset (ReSTIRKernels.cu:36-47)
    hipStream_t st = r->stream;
    for (int i = 0; i < 3; i++) if (hipMemsetAsync(f.gbuf[i], 0, (size_t)8 * n * sizeof(float4), st) != hipSuccess || hipMemsetAsync(f.probe[i], 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    for (int i = 0; i < 5; i++) if (hipMemsetAsync(f.res[i], 0, (size_t)4 * n * sizeof(float4), st) != hipSuccess || hipMemsetAsync(f.resC[i], 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    if (hipMemsetAsync(f.combined, 0, (size_t)n * sizeof(float4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    if (hipMemsetAsync(f.output, 0, (size_t)n * sizeof(uchar4), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "memset failed");
    r->allocN = n;
    r->fenceNeeded = true;
    r->haveEst = false; r->cntPending[0] = r->cntPending[1] = false;
    if (r->dSwap.ensure(1) || hipMemsetAsync(r->dSwap.p, 0, sizeof(int), st) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "swap index allocation failed");
    f.swap = r->dSwap.p;
    r->blendCounter = 0; r->frameIndex = 0; r->gbufIndex = 0; r->lastGbuf = 0;
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
    if (!r->timing) return;
    if (r->evUsed == r->evPool.size()) { R::EvPair p; if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; p.cls = 0; r->evPool.push_back(p); }
    slot = r->evUsed++;
    r->evPool[slot].cls = cls;
    (void)hipEventRecord(r->evPool[slot].a, r->stream);
}
void evEnd(R* r, size_t slot) { if (slot != (size_t)-1) (void)hipEventRecord(r->evPool[slot].b, r->stream); }
void evBegin2(R* r, int cls, size_t& slot, hipStream_t s)
{
    slot = (size_t)-1;
    if (!r->timing) return;
    if (r->evUsed == r->evPool.size()) { R::EvPair p; if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; p.cls = 0; r->evPool.push_back(p); }
    slot = r->evUsed++;
    r->evPool[slot].cls = cls;
    (void)hipEventRecord(r->evPool[slot].a, s);
}
void evEnd2(R* r, size_t slot, hipStream_t s) { if (slot != (size_t)-1) (void)hipEventRecord(r->evPool[slot].b, s); }

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
    {   // scene edits since the last frame go to the device on the stream of the frame front, behind the merge of the frame two
        // back (the last reader of the scene set that is rewritten); see SceneSet
        hipStream_t su = (r->overlap && r->aux != nullptr) ? r->aux : r->stream;
        if (su != r->stream) LM_HIP(hipStreamWaitEvent(su, r->evMerge[r->framePar], 0));
        if ((rc = syncScene(r, su))) return rc;
    }
    if ((rc = ensureFrameBuffers(r))) return rc;
    const LmKernelTable* K = r->K;
    hipStream_t st = r->stream;
    LmFrame& fr = r->fr;
    const uint32_t depthMax = std::min<uint32_t>(r->settings.depth, LM_MAX_DEPTH);
    // "current" / "previous" surface data (the reference toggles two buffers, WaveFrontRenderer.cpp:1045-1049); here three physical
    // sets rotate, so that the next frame's extraction does not wait for this frame's temporal pass
    const int currentIndex = r->gbufIndex, temporalIndex = (r->gbufIndex + 2) % 3;
    const bool blend = r->settings.blend_output != 0;

    // camera (Camera.cpp:79-93,122-140; aspect = render W/H, WaveFrontRenderer.cpp:577)
    LmCamera cam;
    const float aspect = (float)fr.W / (float)fr.H;
    const float halfY = 1.0f * (float)tan((double)(r->fovY * 0.01745329251994329576923690768489f) * 0.5);
    const float halfX = halfY * aspect;
    for (int k = 0; k < 3; k++) { cam.eye[k] = r->camPos[k]; cam.U[k] = r->camRight[k] * halfX; cam.V[k] = r->camUp[k] * halfY; cam.Wv[k] = r->camFwd[k] * 1.0f; }
    float camWorld[16] = {r->camRight[0], r->camUp[0], r->camFwd[0], r->camPos[0], r->camRight[1], r->camUp[1], r->camFwd[1], r->camPos[1],
                          r->camRight[2], r->camUp[2], r->camFwd[2], r->camPos[2], 0, 0, 0, 1};
    if (!r->havePrev) { memcpy(r->prevCamWorld, camWorld, sizeof camWorld); r->havePrev = true; }
    {   // M = projection(fovY, aspect, 0.5, 10000) * inverse(previous camera world matrix)   (WaveFrontRenderer.cpp:763-776)
        float proj[16] = {0}, invPrev[16];
        const float tanHalf = (float)tan((double)(r->fovY * 0.01745329251994329576923690768489f) / 2.0);
        const float zn = 0.5f, zf = 10000.f;
        proj[0] = 1.0f / (aspect * tanHalf); proj[5] = 1.0f / tanHalf;
        proj[10] = -(zf + zn) / (zf - zn); proj[11] = -(2.0f * zf * zn) / (zf - zn); proj[14] = -1.0f;
        invert4(r->prevCamWorld, invPrev);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0.f; for (int k = 0; k < 4; k++) s += proj[i * 4 + k] * invPrev[k * 4 + j]; cam.prevViewProj[i * 4 + j] = s; }
    }

    // ---- frame graph.  Streams: main `st` (ReSTIR chain, merge), `sx` (frame front + indirect waves), aux2 (NEE shadow
    // rays), aux3 (second ReSTIR visibility pass).  Frames are software-pipelined: the front of frame i+1 (primary rays,
    // first closest-hit launch, surface extraction, first continuation) is queued on `sx` behind the waves of frame i and
    // runs beside the ReSTIR tail of frame i on `st`.  What that needs: DIRECT / INDIRECT and the counter block are
    // double-buffered by frame parity; extraction waits for frame i's temporal pass (the last reader of the G-buffer /
    // probe plane / motion vectors it overwrites); a frame's front waits for the merge of the frame two back (owner of
    // the same parity buffers).  Accumulation order per pixel is unchanged, so results equal the serial order bit for bit.
    const bool overlap = r->overlap && r->aux != nullptr;
    hipStream_t sx = overlap ? r->aux : st;
    const int par = r->framePar; r->framePar ^= 1;
    fr.motion = r->dMotion[par].p;
    fr.direct = r->dDirect[par].p; fr.indirect = r->dIndirect[par].p; fr.counters = r->dCounters.p + (size_t)par * LM_CNT_WORDS;
    if (overlap) {
        if (r->fenceNeeded) { LM_HIP(hipEventRecord(r->evTop, st)); LM_HIP(hipStreamWaitEvent(sx, r->evTop, 0)); }
        LM_HIP(hipStreamWaitEvent(sx, r->evMerge[par], 0));
    }
    r->fenceNeeded = false;
    size_t evAll; evBegin2(r, 4, evAll, sx);
    LM_HIP(hipMemsetAsync(fr.counters, 0, LM_CNT_WORDS * sizeof(uint32_t), sx));
    if (!blend) K->clear(st, r->gridFor(fr.n, 8), fr.combined, fr.n);                        // :559
    ++r->frameCount;                                                                          // :593
    K->primary(sx, r->gridFor(fr.n, 8), fr, cam, r->frameCount);
    uint32_t seed = wangHash(r->frameCount);                                                  // :685
    LmScene scx = r->dscene;                                                                  // same scene, its own stack-spill area
    if (overlap) scx.spill += (size_t)r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
    const int gridMain = r->numCU * r->traceBlocksMain, gridAux = r->numCU * (overlap ? r->traceBlocksAux : r->traceBlocksMain);
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
    const uint32_t tailBelow = r->tailBelow >= 0 ? (uint32_t)r->tailBelow : (fr.n < (1u << 20) ? 65536u : 16384u);
    if (tailBelow && r->haveEst) for (uint32_t dd = 1; dd < depthMax; dd++) if (r->estRays[dd] < tailBelow) { tailDepth = (int)dd; break; }
    int q = 0;
    size_t ev;
    bool tailLaunched = false;
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
            K->trace_closest(sx, gridMain, scx, fr.rayO[q], fr.rayD[q], inCount, fr.hits, 0.01f, 5000.f, fr.counters, r->refillPrimary);    // :678,:703
            evEnd2(r, ev, sx);
            if (overlap) LM_HIP(hipStreamWaitEvent(sx, r->evTemporal[par], 0));          // the temporal pass two frames back has read what extraction overwrites
            evBegin2(r, 2, ev, sx);
            K->extract0(sx, r->gridFor(fr.n, 8), r->dscene, (int)depth + 1 == tailDepth ? withTailQueue(fr, q ^ 1) : fr, cam, currentIndex, seed2, doIndirect, q ^ 1, outCount);   // + depth-0 continuation
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
            evBegin2(r, 3, ev, sp);
            const int cur = LM_RES_CUR, tmp = LM_RES_PREV, fresh = pickAhead ? 4 : LM_RES_CUR;
            uint32_t rs = wangHash(seed);
            K->fill_bags(sp, r->dscene, fr, seed, 50u * 1000u);
            rs = wangHash(rs);
            const uint32_t tx0 = fr.x0 / 16u, ty0 = fr.y0 / 16u;
            const uint32_t wtx = (fr.x0 + fr.ww + 15u) / 16u - tx0, wty = (fr.y0 + fr.wh + 15u) / 16u - ty0;
            K->pick_primary(sp, (int)(wtx * wty), r->dscene, fr, currentIndex, fresh, rs, fr.counters + LM_CNT_RESTIR(0));   // + visibility rays, pass 1
            LmScene scp = r->dscene;                                 // the pick-ahead stream traces with its own stack-spill area
            if (sp != st) scp.spill += (size_t)3 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
            K->trace_shade(sp, gridMain, scp, fr, fresh, fr.counters + LM_CNT_RESTIR(0), r->refillVisibility, 0);
            evEnd2(r, ev, sp);
            if (pickAhead) { LM_HIP(hipEventRecord(r->evPick, sp)); LM_HIP(hipStreamWaitEvent(st, r->evPick, 0)); }
            evBegin(r, 3, ev);
            rs = wangHash(rs);
            K->temporal(st, tiles, fr, currentIndex, temporalIndex, cur, tmp, fresh, rs, fr.counters + LM_CNT_RESTIR(1));         // + visibility rays, pass 2
            if (overlap) LM_HIP(hipEventRecord(r->evTemporal[par], st));
            rs = wangHash(rs);
            K->spatial(st, tiles, fr, currentIndex, cur, 2, rs, 30);
            // second visibility pass (ReSTIR.cpp:211-212) works on the CURRENT buffer, which the second spatial pass does not
            // touch: trace it beside that pass.  (It must follow the first spatial pass, which reads the current buffer.)
            hipStream_t sv = (overlap && !pickAhead) ? r->aux3 : st;
            LmScene scv = r->dscene;
            if (sv != st) {
                scv.spill += (size_t)3 * r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
                LM_HIP(hipEventRecord(r->evVis, st)); LM_HIP(hipStreamWaitEvent(sv, r->evVis, 0));
            }
            K->trace_shade(sv, gridMain, scv, fr, cur, fr.counters + LM_CNT_RESTIR(1), r->refillVisibility, 1);
            if (sv != st) LM_HIP(hipEventRecord(r->evVisDone, sv));
            K->spatial(st, tiles, fr, currentIndex, 2, 3, rs, 0);
            if (sv != st) LM_HIP(hipStreamWaitEvent(st, r->evVisDone, 0));
            K->combine(st, tiles, fr, currentIndex, cur, 3, wangHash(rs));
            evEnd(r, ev);
        } else if ((int)depth >= tailDepth) {
            // path tail: the remaining waves in one launch (kernels.hip lm_k_path_tail) on the shadow stream: its INDIRECT adds
            // follow the previous wave's NEE adds by stream order, and the wave stream is free for the next frame's front
            hipStream_t stl = overlap ? r->aux2 : sx;
            LmScene sct = scx;
            if (overlap) {
                sct.spill += (size_t)r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS);
                LM_HIP(hipEventRecord(r->evShade[depth], sx)); LM_HIP(hipStreamWaitEvent(stl, r->evShade[depth], 0));     // the queue's producer is done
            }
            evBegin2(r, 0, ev, stl);
            K->path_tail(stl, r->numCU * 8, sct, withTailQueue(fr, q), q, inCount, (int)depth, (int)depthMax, seed, r->tailLanes);
            evEnd2(r, ev, stl);
            if (overlap) LM_HIP(hipEventRecord(r->evTail, stl));
            tailLaunched = true;
            break;
        } else {
            uint32_t* shCount = fr.counters + LM_CNT_SHADOW(depth);
            evBegin2(r, 0, ev, sx);
            K->trace_closest(sx, gridAux, scx, fr.rayO[q], fr.rayD[q], inCount, fr.hits, 0.01f, 5000.f, fr.counters, r->refillBelow);
            evEnd2(r, ev, sx);
            if (overlap) LM_HIP(hipStreamWaitEvent(sx, r->evJoin2, 0));                   // previous wave's (or frame's) shadow rays consumed
            evBegin2(r, 2, ev, sx);
            K->shade_wave(sx, r->numCU * 8, scx, (int)depth + 1 == tailDepth ? withTailQueue(fr, q ^ 1) : fr, q, inCount, seed, seed2, doIndirect, outCount, shCount);
            evEnd2(r, ev, sx);
            // NEE shadow rays of this wave: third stream, beside the next wave's closest-hit launch.  The shadow queue is
            // rewritten by the NEXT shade_wave, which therefore waits for this launch (evJoin2).  (`shadow_on_wave` 1 keeps them on
            // the wave stream: equal at full size and for the windows of 4 / 8 ranks, 8 % slower for those of 2 ranks.)
            const bool shadowOnWave = r->shadowOnWave != 0;
            hipStream_t ss = (overlap && !shadowOnWave) ? r->aux2 : sx;
            LmScene scs = scx;
            if (ss != sx) { scs.spill += (size_t)r->traceGrid() * 256 * (LM_STACK_DEPTH - LM_STACK_LDS); LM_HIP(hipEventRecord(r->evShade[depth], sx)); LM_HIP(hipStreamWaitEvent(ss, r->evShade[depth], 0)); }
            evBegin2(r, 1, ev, ss);
            K->trace_shadow(ss, gridAux, scs, fr, shCount, 0.01f, r->refillBelow);     // tmin of the intersection launch (:843)
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
    K->merge(st, r->gridFor(fr.n, 8), fr, blend ? 1 : 0, r->blendCounter, (int)depthMax);     // + ReSTIR::SwapBuffers per executed wave
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
        r->frameStats["Wavefront Iteration"] = (uint64_t)((r->classMs[0] + r->classMs[2] + r->classMs[3]) * 1000.f / frames);
        r->frameStats["Shadow Rays"] = (uint64_t)(r->classMs[1] * 1000.f / frames);
        r->frameStats["ReSTIR"] = (uint64_t)(r->classMs[3] * 1000.f / frames);
        r->frameStats["Total Frame Time"] = (uint64_t)(r->classMs[4] * 1000.f / frames);
    }
    return 0;
}

}  // namespace

// ==============================================================================================================
// C ABI
// ==============================================================================================================
extern "C" {

const char* lumen_mi_last_error(void) { return g_lastError.c_str(); }

int lumen_mi_create(lumen_mi_renderer** out)
{
    if (!out) return fail(LUMEN_MI_ERR_INVALID, "out is NULL");
    initLut();
    *out = new lumen_mi_renderer();
    (*out)->K = lm_kernel_table();
    if (const char* e = getenv("LUMEN_MI_REFILL")) (*out)->refillBelow = atoi(e);
    if (const char* e = getenv("LUMEN_MI_REFILL_VIS")) (*out)->refillVisibility = atoi(e);
    if (const char* e = getenv("LUMEN_MI_REFILL_PRIMARY")) (*out)->refillPrimary = atoi(e);
    if (const char* e = getenv("LUMEN_MI_TAIL_BELOW")) (*out)->tailBelow = atoi(e);
    if (const char* e = getenv("LUMEN_MI_PICK_AHEAD")) (*out)->pickAhead = atoi(e);
    if (const char* e = getenv("LUMEN_MI_SHADOW_ON_WAVE")) (*out)->shadowOnWave = atoi(e);
    if (const char* e = getenv("LUMEN_MI_TAIL_LANES")) (*out)->tailLanes = std::max(1, std::min(64, atoi(e)));
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
    if (const char* e = getenv("LUMEN_MI_TRACE_BLOCKS_MAIN")) r->traceBlocksMain = std::max(1, std::min(8, atoi(e)));
    if (const char* e = getenv("LUMEN_MI_TRACE_BLOCKS_AUX")) r->traceBlocksAux = std::max(1, std::min(8, atoi(e)));
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
        if (r->aux) { (void)hipStreamSynchronize(r->aux); (void)hipStreamSynchronize(r->aux2); (void)hipStreamDestroy(r->aux2); (void)hipStreamSynchronize(r->aux3); (void)hipStreamDestroy(r->aux3); (void)hipEventDestroy(r->evPick); (void)hipEventDestroy(r->evVis); (void)hipEventDestroy(r->evVisDone); (void)hipEventDestroy(r->evJoin2); for (auto& e : r->evShade) (void)hipEventDestroy(e); (void)hipStreamDestroy(r->aux); (void)hipEventDestroy(r->evFront); for (auto& e : r->evTemporal) (void)hipEventDestroy(e); (void)hipEventDestroy(r->evTail); (void)hipEventDestroy(r->evTop); for (auto& e : r->evMerge) (void)hipEventDestroy(e); for (int i = 0; i < 2; i++) { (void)hipEventDestroy(r->evCnt[i]); (void)hipHostFree(r->pinnedCounters[i]); r->pinnedCounters[i] = nullptr; } (void)hipEventDestroy(r->evJoin); }
        for (SceneSet& S : r->sset) { if (S.upPending) (void)hipEventSynchronize(S.evUp); S.release(); }
        r->dSpill.release(); r->dTriId.release(); r->dTriOrder.release(); r->dVerts.release(); r->dIndices.release();
        r->dTriBox.release(); r->dNodeBox.release(); r->dLevelNodes.release(); r->dRefitBounds.release();
        r->dMaterials.release(); r->dTexDesc.release(); r->dTexels.release(); r->dLut.release();
        for (auto& b : r->dRay) b.release(); for (auto& b : r->dTailRay) b.release(); for (auto& b : r->dSh) b.release(); for (auto& b : r->dSh2) b.release(); for (auto& b : r->dGbuf) b.release(); for (auto& b : r->dProbe) b.release(); for (auto& b : r->dRes) b.release(); for (auto& b : r->dResC) b.release();
        for (int i = 0; i < 2; i++) { r->dDirect[i].release(); r->dIndirect[i].release(); } r->dCombined.release(); r->dHits.release(); for (auto& b : r->dMotion) b.release(); r->dCounters.release(); r->dSwap.release(); r->dOutput.release(); r->dBags.release();
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
    r->textures.push_back(std::move(t));
    r->texturesDirty = true;
    *out = mkh(H_TEXTURE, r->textures.size() - 1);
    return 0;
}

int lumen_mi_create_default_resources(lumen_mi_renderer* r, lumen_mi_handle* white, lumen_mi_handle* normal, lumen_mi_handle* diffuse)
{
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
    r->materialsDirty = true;
    *out = mkh(H_MATERIAL, r->materials.size() - 1);
    return 0;
}

int lumen_mi_update_material(lumen_mi_renderer* r, lumen_mi_handle material, const lumen_mi_material_data* d)
{
    size_t idx;
    if (!r || !d || !unh(material, H_MATERIAL, r->materials.size(), idx)) return fail(LUMEN_MI_ERR_INVALID, "bad material handle");
    Material m;
    const int rc = fillMaterial(r, d, m);
    if (rc) return rc;
    ApiLock lk(r);
    r->materials[idx] = m;
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
    if (d->interleaved) {
        if (!d->vertex_binary) return fail(LUMEN_MI_ERR_INVALID, "interleaved primitive without vertex_binary");
        memcpy(p.verts.data(), d->vertex_binary, (size_t)d->n_vertices * 48);
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
    size_t s;
    if (!r || !unh(scene, H_SCENE, r->scenes.size(), s)) return fail(LUMEN_MI_ERR_INVALID, "bad scene handle");
    ApiLock lk(r);
    r->activeScene = (long)s; r->sceneDirty = true;
    return 0;
}

int lumen_mi_scene_add_mesh(lumen_mi_renderer* r, lumen_mi_handle scene, lumen_mi_handle mesh, lumen_mi_handle* inst)
{
    size_t s, m;
    if (!r || !inst || !unh(scene, H_SCENE, r->scenes.size(), s) || !unh(mesh, H_MESH, r->meshes.size(), m)) return fail(LUMEN_MI_ERR_INVALID, "bad scene/mesh handle");
    ApiLock lk(r);
    Instance i;
    i.scene = s; i.mesh = m;
    const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(i.M, id, sizeof id);
    i.mode = LUMEN_MI_EMISSION_ENABLED; i.radiance[0] = i.radiance[1] = i.radiance[2] = 0.f; i.scale = 1.f; i.overrideMaterial = -1;   // MeshInstance.h:24-35
    r->instances.push_back(i);
    r->scenes[s].instances.push_back(r->instances.size() - 1);
    r->sceneDirty = true;
    *inst = mkh(H_INSTANCE, r->instances.size() - 1);
    return 0;
}

int lumen_mi_scene_clear(lumen_mi_renderer* r, lumen_mi_handle scene)
{
    size_t s;
    if (!r || !unh(scene, H_SCENE, r->scenes.size(), s)) return fail(LUMEN_MI_ERR_INVALID, "bad scene handle");
    ApiLock lk(r);
    r->scenes[s].instances.clear(); r->sceneDirty = true;
    return 0;
}

int lumen_mi_instance_set_transform(lumen_mi_renderer* r, lumen_mi_handle inst, const float m[16])
{
    size_t i;
    if (!r || !m || !unh(inst, H_INSTANCE, r->instances.size(), i)) return fail(LUMEN_MI_ERR_INVALID, "bad instance handle");
    ApiLock lk(r);
    if (memcmp(r->instances[i].M, m, 64) != 0) { memcpy(r->instances[i].M, m, 64); r->transformsDirty = true; }   // polled every frame by the adapter
    return 0;
}

int lumen_mi_instance_set_emissiveness(lumen_mi_renderer* r, lumen_mi_handle inst, int mode, const float rad[3], float scale)
{
    size_t i;
    if (!r || !rad || mode < 0 || mode > 2 || !unh(inst, H_INSTANCE, r->instances.size(), i)) return fail(LUMEN_MI_ERR_INVALID, "bad emissiveness arguments");
    ApiLock lk(r);
    Instance& x = r->instances[i];
    if (x.mode != mode || x.radiance[0] != rad[0] || x.radiance[1] != rad[1] || x.radiance[2] != rad[2] || x.scale != scale) r->entriesDirty = true;
    x.mode = mode; x.radiance[0] = rad[0]; x.radiance[1] = rad[1]; x.radiance[2] = rad[2]; x.scale = scale;
    return 0;
}

int lumen_mi_instance_set_override_material(lumen_mi_renderer* r, lumen_mi_handle inst, lumen_mi_handle mat)
{
    size_t i, m;
    if (!r || !unh(inst, H_INSTANCE, r->instances.size(), i) || !unh(mat, H_MATERIAL, r->materials.size(), m)) return fail(LUMEN_MI_ERR_INVALID, "bad handle");
    ApiLock lk(r);
    if (r->instances[i].overrideMaterial != (long)m) r->entriesDirty = true;
    r->instances[i].overrideMaterial = (long)m;
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
int lumen_mi_get_render_resolution(lumen_mi_renderer* r, uint32_t* w, uint32_t* h) { if (!r || !w || !h) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); *w = r->pending.render_width; *h = r->pending.render_height; return 0; }
int lumen_mi_get_output_resolution(lumen_mi_renderer* r, uint32_t* w, uint32_t* h) { if (!r || !w || !h) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); *w = r->pending.output_width; *h = r->pending.output_height; return 0; }
int lumen_mi_set_blend_mode(lumen_mi_renderer* r, int b) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); r->pending.blend_output = b ? 1 : 0; if (b) r->blendCounter = 0; return 0; }
int lumen_mi_get_blend_mode(lumen_mi_renderer* r, int* b) { if (!r || !b) return fail(LUMEN_MI_ERR_INVALID, "NULL argument"); *b = r->pending.blend_output; return 0; }
int lumen_mi_set_depth(lumen_mi_renderer* r, uint32_t d) { if (!r || d == 0 || d > LM_MAX_DEPTH) return fail(LUMEN_MI_ERR_INVALID, "depth must be in [1, 16]"); r->pending.depth = d; return 0; }

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

static int copyOut(lumen_mi_renderer* r, const void* dev, size_t bytes, void* host, size_t capacity)
{
    if (!r || !host) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (!dev) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    if (capacity < bytes) return fail(LUMEN_MI_ERR_INVALID, "buffer too small");
    ApiLock lk(r);
    int rc = syncAndCollect(r);
    if (rc) return rc;
    LM_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int lumen_mi_get_output_pixels(lumen_mi_renderer* r, uint8_t* rgba8, size_t cap, uint32_t* w, uint32_t* h)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    if (w) *w = r->fr.ww; if (h) *h = r->fr.wh;
    return copyOut(r, r->fr.output, (size_t)r->fr.n * 4, rgba8, cap);
}
int lumen_mi_get_radiance(lumen_mi_renderer* r, float* out, size_t cap) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); return copyOut(r, r->fr.combined, (size_t)r->fr.n * 16, out, cap); }
int lumen_mi_get_channel(lumen_mi_renderer* r, int ch, float* out, size_t cap)
{
    if (!r || ch < 0 || ch > 1) return fail(LUMEN_MI_ERR_INVALID, "channel must be 0 (DIRECT) or 1 (INDIRECT)");
    return copyOut(r, ch == 0 ? r->fr.direct : r->fr.indirect, (size_t)r->fr.n * 16, out, cap);
}
int lumen_mi_copy_radiance_device(lumen_mi_renderer* r, void* dst)
{
    if (!r || !dst) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    if (!r->fr.combined) return fail(LUMEN_MI_ERR_STATE, "no frame has been traced yet");
    LM_HIP(hipMemcpyAsync(dst, r->fr.combined, (size_t)r->fr.n * 16, hipMemcpyDeviceToDevice, r->stream));
    return 0;
}
int lumen_mi_get_gbuffer(lumen_mi_renderer* r, float* out, size_t cap)
{
    if (!r || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    const uint32_t n = r->fr.n;
    if (cap < (size_t)n * 128) return fail(LUMEN_MI_ERR_INVALID, "buffer too small");
    const int last = r->lastGbuf;
    return copyOut(r, r->fr.gbuf[last], (size_t)n * 128, out, cap);
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
    for (uint32_t i = 0; i < n && i < 64; i++) out[i] = v[i];
    return 0;
}
int lumen_mi_get_kernel_time(lumen_mi_renderer* r, int which, float* ms, uint32_t* launches)
{
    if (!r || which < 0 || which > 4) return fail(LUMEN_MI_ERR_INVALID, "bad kernel class");
    if (ms) *ms = r->classMs[which]; if (launches) *launches = r->classLaunches[which];
    return 0;
}
int lumen_mi_enable_kernel_timing(lumen_mi_renderer* r, int e)
{
    if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer");
    ApiLock lk(r);
    if (e) { for (int c = 0; c < 5; c++) { r->classMs[c] = 0.f; r->classLaunches[c] = 0; } }      // enabling starts a new accumulation window
    r->timing = e != 0;
    return 0;
}
int lumen_mi_set_tuning(lumen_mi_renderer* r, const char* key, int value)
{
    if (!r || !key) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    const std::string k = key;
    if (k == "tail_below") r->tailBelow = value;
    else if (k == "tail_lanes") r->tailLanes = std::max(1, std::min(64, value));
    else if (k == "single_stream") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->overlap = value == 0; }
    else if (k == "refit") r->refitEnabled = value;
    else if (k == "pick_ahead") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->pickAhead = value; }
    else if (k == "shadow_on_wave") { if (r->initialised) { int rc = syncAndCollect(r); if (rc) return rc; } r->shadowOnWave = value; }
    else if (k == "refill") r->refillBelow = value;
    else if (k == "refill_visibility") r->refillVisibility = value;
    else if (k == "refill_primary") r->refillPrimary = value;
    else return fail(LUMEN_MI_ERR_INVALID, std::string("unknown tuning key: ") + key);
    return 0;
}
int lumen_mi_set_instrumented(lumen_mi_renderer* r, int e) { if (!r) return fail(LUMEN_MI_ERR_INVALID, "NULL renderer"); ApiLock lk(r); r->instrumented = e != 0; r->K = e ? lm_kernel_table_instrumented() : lm_kernel_table(); return 0; }

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
    r->K->query_closest(r->stream, r->traceGrid(), r->dscene, dO.p, dD.p, n, tmin, tmax, dI.p, dU.p, r->dCounters.p);
    std::vector<uint4> hi(n); std::vector<float4> hu(n);
    LM_HIP(hipStreamSynchronize(r->stream));
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

int lumen_mi_test_bsdf(lumen_mi_renderer* r, uint32_t n, int mode, const float* mat23, const float* N, const float* T, const float* wo, const float* aux, float* out8)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!mat23 || !N || !T || !wo || !aux || !out8) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
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
int lumen_mi_test_math(lumen_mi_renderer* r, uint32_t n, int fn, const float* x, const float* y, float* out)
{
    if (!r || !r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised");
    if (!x || !y || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
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
int lumen_mi_get_world_triangles(lumen_mi_renderer* r, float* out, uint32_t cap, uint32_t* count)
{
    if (!r || !count) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    ApiLock lk(r);
    int rc = prepareScene(r); if (rc) return rc;
    *count = (uint32_t)r->triEntry.size();
    if (out) { if (cap < *count) return fail(LUMEN_MI_ERR_INVALID, "buffer too small"); memcpy(out, r->worldTris.data(), r->worldTris.size() * 4); }
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
    if (nodes) *nodes = (uint32_t)r->bvh.nodes.size(); if (tris) *tris = (uint32_t)r->bvh.order.size(); if (maxDepth) *maxDepth = r->bvh.maxDepth;
    return 0;
}

}  // extern "C"
