// renderer_state.h — internal to liblumen_mi.so: the renderer object behind the opaque `lumen_mi_renderer*` of
// include/lumen_mi.h, its host-side resource tables and device buffers, and the functions the three host translation units
// share (scene.cpp: flattening, BVH / scene-set upload, light list; frame.cpp: frame buffers and the frame graph;
// renderer.cpp: the C ABI).  Not installed; include/lumen_mi.h is the only public header.
#pragma once
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
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void lm_read_pushes(hipStream_t s, unsigned long long* out);

namespace lmr {

inline thread_local std::string g_lastError;
inline int fail(int code, const std::string& msg) { g_lastError = msg; return code; }

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

struct Texture { uint32_t w, h; bool srgb; std::vector<uint32_t> px; uint8_t minG = 255; };      // minG: smallest green texel (roughness channel of a metal-roughness map)
struct Material {
    LmDevMaterial dev; float emissiveColor[3];
    bool mayBeRare = false;     // can a surface of this material fall outside the contracted ReSTIR evaluation (lm_bsdf.h lm_quick_contracts)?  dielectric
                                // or clear-coat factor > 0, or a roughness byte that can truncate to 0 given the smallest roughness texel
};
struct Primitive { std::vector<Vertex48> verts; std::vector<uint32_t> idx; size_t material; std::vector<uint8_t> emissive; uint32_t numLights = 0; bool containEmissive = false; };
struct Mesh {
    std::vector<size_t> prims;
    // cached for instance-level assembly (scene.cpp flatten): the mesh's own tree over its object-space triangles, its box
    std::shared_ptr<LmBvh> bvh; float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; uint32_t tris = 0;
};
struct Instance {
    size_t scene; size_t mesh; float M[16]; int mode; float radiance[3]; float scale; long overrideMaterial; std::vector<uint32_t> entries;
    uint32_t gen = 1; bool alive = true;     // slots of a cleared scene are reused; the handle carries the generation, so a stale handle is refused
};
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
    DevBuf<LmNodeW> top;                                                     // top-of-tree table of `nodes` (lm_k_build_top)
    DevBuf<LmNodeW> nodes; DevBuf<LmTriPacket> packets; DevBuf<float> quant; DevBuf<LmEntry> entries; DevBuf<LmLight> lights; DevBuf<float> cdf;
    DevBuf<uint2> triId; DevBuf<uint32_t> triOrder, levelNodes;            // topology of the tree in `nodes` (what an instance add / remove rewrites)
    std::vector<uint32_t> levelStart; uint32_t nTris = 0;
    HostBuf<LmEntry> hEntries; HostBuf<LmLight> hLights; HostBuf<float> hCdf;
    HostBuf<LmNodeW> hNodes; HostBuf<uint2> hTriId; HostBuf<uint32_t> hOrder, hLevelNodes;
    hipEvent_t evUp = nullptr; bool upPending = false;      // the staging buffers are free again once this event has passed
    uint64_t entriesVer = 0, geomVer = 0, lightsVer = 0, topoVer = 0;    // state of the host scene this set holds
    void release() {
        top.release(); nodes.release(); packets.release(); quant.release(); entries.release(); lights.release(); cdf.release();
        triId.release(); triOrder.release(); levelNodes.release();
        hEntries.release(); hLights.release(); hCdf.release(); hNodes.release(); hTriId.release(); hOrder.release(); hLevelNodes.release();
        if (evUp) { (void)hipEventDestroy(evUp); evUp = nullptr; }
    }
};

void texel(const Texture& t, int x, int y, float out[4]);      // scene.cpp: host texture fetch, same definition as the device fetch
inline float g_srgbLut[256];
inline void initLut() { static bool d = false; if (d) return; for (int i = 0; i < 256; i++) { const double c = i / 255.0; g_srgbLut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4)); } d = true; }

inline uint32_t wangHash(uint32_t s) { s = (s ^ 61u) ^ (s >> 16); s *= 9u; s = s ^ (s >> 4); s *= 0x27d4eb2du; s = s ^ (s >> 15); return s; }
inline void pack8(uint32_t& w, uint32_t shift, float v) { const uint32_t q = (uint32_t)(v * 255.f); w &= ~(255u << shift); w |= q << shift; }

inline void mulPoint(const float* m, const float* v, float w, float* out)     // rows 0..2, operation order of sutil Matrix4x4 * float4
{
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3] * w;
    out[1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7] * w;
    out[2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11] * w;
}

}  // namespace lmr
using namespace lmr;            // internal header: only the host translation units of this library include it

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
    int assembleEnabled = 1;                // topology edits after the first build assemble cached per-mesh trees + GPU refit instead of a host SAH rebuild
    bool builtOnce = false;
    uint32_t assemblies = 0;                // scene trees assembled since creation
    uint32_t fuzz = 0;                      // != 0: schedule fuzzing (test aid): random idle launches in front of the kernels of a frame, this is the RNG state
    int shadowOnWave = 0;                   // 1: NEE shadow rays on the wave stream (the path tail then has the third stream to itself); measured: 8 % slower for half-frame windows, equal elsewhere
    hipEvent_t evFront = nullptr, evTemporal[2] = {nullptr, nullptr}, evTop = nullptr, evMerge[2] = {nullptr, nullptr};   // cross-frame pipelining (traceFrameAsync)
    int framePar = 0;                       // parity of the frame being enqueued: selects the channel buffers and the counter block
    int fenceNeeded = 2;                    // main-stream work (uploads, memsets) the frame front must wait for: counts the frames that still have to
                                            // fence (two: with two wave streams each stream's first frame)
    std::vector<hipEvent_t> evShade;        // per wave: shade_wave(d) done
    int auxPriority = 1;                    // 1: highest priority for the aux streams, 0: default
    int aux3Priority = 0;                   // the visibility / pick-ahead stream runs at default priority (pick-ahead must not starve the main chain)
    bool overlap = true;
    int traceBlocksMain = 0, traceBlocksAux = 8, traceBlocksVis = 0;       // blocks per CU of the persistent traversal launches: primary rays, waves >= 1 + shadow rays, visibility passes; 0 = chosen per frame (frame.cpp)
    int numCU = 256;
    const LmKernelTable* K = nullptr;
    LmKernelTable Kmix;                     // the renderer's own table: every entry from the default compilation or, per kernel class, from the -fno-slp-vectorize one (renderer.cpp applyNoSlpKernels)
    bool instrumented = false;
    int tailBelow = -1;                     // waves expected to hold fewer rays than this run as one path-tail launch (0 = off,
                                            // -1 = auto: 65536 for windows under 1 Mpixel, where the wave chain is the critical path, else 16384) ...
    int waveStreams = 1;                    // 1 (default): every frame's waves on one stream, NEE shadows + path tail on another; 2: the path-tracing work of
                                            // even / odd frames alternates between those two streams (each frame's closest-hit, shading, shadow and tail
                                            // launches in series on its own), so that consecutive frames' wave chains overlap.  Measured box-dependent:
                                            // + 1.5 % on one box, - 3 % on two others (tiles - 6 ... - 8 %), toy frames + 24 %: off by default
    int tailPair = -1;                      // path tail in pair mode (shadow ray of depth d traced by a partner lane beside the closest-hit query of depth d + 1):
                                            // 1 on (at most 32 paths per wavefront), 0 off, -1 automatic: on for windows under 1.5 Mpixel
    int tailGrid = 8;                       // blocks per CU of the path-tail launch (environment LUMEN_MI_TAIL_GRID: A/B knob, profiles/r06_tail_grid_ab.txt)
    int tailLanes = -1;                     // ... with this many paths per wavefront (-1 = auto: 64 for windows from 0.75 Mpixel, where the tail hides behind
                                            // the other streams and fuller wavefronts save VALU issue slots; 16 for smaller windows, where the tail IS the critical path)
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
    int texFilter = 0;              // 0: CUDA's published linear-filter rule (1.8 fixed-point weights), 1: unquantised fp32 weights (tuning key tex_filter; decision D6)
    bool transformsDirty = false;           // only instance matrices changed since the last build: the BVH is refitted on the GPU
    bool entriesDirty = false;              // emissive mode / radiance / override material of an instance changed: scene table + lights only
    uint32_t refits = 0;                    // refits since the last full build
    int refitEnabled = 1;                   // 0: every transform change triggers a full host rebuild
    bool anyRareMaterial = false;           // some material ever created may need the second (exact) launch of the fast ReSTIR passes
    int fastShade = 0;                      // the NEE contribution of the shading kernels (depth >= 1) in the fast arithmetic policy (lm_shade.h lm_shade_direct): changes radiance
                                            // in the last bits only; sampling / Russian roulette stay exact.  Off: measured gain below 2 % (profiles/r03_fast_shade_ab.txt)
    // Lazy reuse (tuning key lazy_reuse; frame.cpp): the spatial passes and the combine of a frame are not launched with the frame but at the start of the
    // next frame's ReSTIR chain, where the device runs them only if their result can still be read (kernels.hip lm_reuse_owed).  1 on, 0 off (launched with
    // their frame, every frame), -1 automatic: on at even path depths, where the reference's swap quirk makes every such result dead while the camera rests.
    int fusePrimary = 0;                    // 1: primary rays generated inside the packet traversal of the primary wave (tuning key fuse_primary) instead of their own launch first.
                                            // Same image; measured - 0.3 % (profiles/r03_fuse_primary_ab.txt): the launch it saves only waited for slots the traversal then waits for
    int lazyReuse = -1;
    struct OwedReuse { bool valid = false; LmFrame fr{}; int gbuf = 0; uint32_t seed = 0; int fast = 0; int tiles = 0; } owed;
    std::atomic<uint64_t> framesTraced{0};   // TraceFrames enqueued since creation (frame-stat key "Frames Traced")
    int gpuBuild = 0;                       // 1: a scene's full tree build runs on the device (bvh_gpu.hip, LBVH); 0: host binned-SAH build (better tree, default)
    uint64_t gpuBuilds = 0;                 // device builds since creation (counter [56])
    int tailRepack = 0;                     // path tail: 1 = the repacking variant (256 paths per block, survivors packed through LDS after every depth); 0 = one path per lane to the end
    int fuseCombine = 1;                    // eager frames: the second spatial pass ends with the combine (lm_k_restir_spatial*_fused): 1 on (default), 0 off
    int pickWide = 1;                       // light lists of 513 .. 1 984 records: the candidate pick as 1024-thread blocks around one LDS table: 0 never, 1 fast mode only, 2 both modes
    int spatialLds = 0;                     // fast mode: the first spatial pass stages its probe window in LDS (lm_k_restir_spatial_fast_lds): 1 on, 0 off
    int packetVisibility = 0;               // the ReSTIR visibility rays likewise (lm_k_restir_trace_shade_packet): 1 on, 0 off (default), -1 the primary wave's rule.
                                            // Measured 3x SLOWER on C2 (profiles/r03_packet_visibility_ab.txt): a tile's visibility rays start on surfaces at very
                                            // different depths, the union of their paths is large, and an any-hit packet runs until its last unoccluded ray is through
    int packetPrimary = -1;                 // the primary wave is traced as packets (one shared stack per wavefront, lm_trace_packets): 1 on, 0 off,
                                            // -1 auto = when the window has more than 3 pixels per scene triangle (a wavefront's 8 x 8 pixel tile then meets
                                            // few distinct leaves: C2 / C3 +1.2 % at 14 pixels per triangle, the Sandbox's 720p window +0.85 % at 3.5, profiles/
                                            // r04_sandbox_packet_ab.txt; with sub-pixel geometry the union of 64 rays' nodes costs more: C5 -11 % at 0.2)
    int sortRays = 0;                       // > 0: continuation-ray queues of waves 1 .. sortRays are reordered by (origin cell, octant) before their closest-hit launch
    DevBuf<uint32_t> dSortBins;             // 2 x 4096 words (histogram + cursors)
    int fastResample = 0;                   // 1: the ReSTIR target function and resampling weights use hardware rcp / rsq / sqrt (LmFast): within 1e-3 rel-L2
                                            // of the exact mode, not bit-identical (tuning key "fast_resample", LUMEN_MI_FAST_RESAMPLE)

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
    DevBuf<uint32_t> dHazard[2];            // lazy reuse: LmFrame::hazardList, by frame parity
    DevBuf<int> dSwap;                      // ReSTIR swap-chain index lives on the device (LmFrame::swap)

    // flattened scene (host)
    std::vector<LmEntry> entries;
    std::vector<size_t> freeInstances;      // instance slots released by lumen_mi_scene_clear
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
    uint64_t entriesVer = 1, geomVer = 1, lightsVer = 1, topoVer = 1;      // versions of the host-side scene state (instance table, geometry, light list, tree topology)
    std::vector<uint2> triId;               // per BVH triangle slot: (table entry, primitive-local triangle)
    std::vector<uint32_t> vertBase, idxBase; size_t poolPrims = 0, poolVerts = 0, poolIdx = 0;      // vertex / index pools on the device cover prims [0, poolPrims)
    DevBuf<float4> dVerts; DevBuf<uint32_t> dIndices; DevBuf<LmDevMaterial> dMaterials;
    DevBuf<float4> dTriBox, dNodeBox; DevBuf<uint32_t> dRefitBounds;
    DevBuf<int> dSpill; DevBuf<LmTexDesc> dTexDesc; DevBuf<uint32_t> dTexels; DevBuf<float> dLut;
    LmScene dscene{};

    // device frame
    LmFrame fr{};
    uint32_t allocN = 0, allocDepth = 0;
    DevBuf<float4> dTailRay[6];             // ray queue of the path tail, double-buffered by frame parity (3 planes each)
    hipEvent_t evTail = nullptr, evScene = nullptr;
    DevBuf<float4> dRay[12], dSh[6], dSh2[4];      // ray queues and the NEE shadow queue once per frame parity (two wave streams: consecutive frames trace concurrently)
    DevBuf<float4> dGbuf[3], dProbe[3], dRes[5], dResC[5], dDirect[2], dIndirect[2], dCombined;
    DevBuf<uint4> dHits[2]; DevBuf<uint32_t> dMotion[2], dCounters, dReuseMask, dRareTile[3]; DevBuf<uchar4> dOutput; DevBuf<uint2> dBags;
    DevBuf<uint2> dExportHalf;                          // lumen_mi_get_radiance_half4
    DevBuf<unsigned long long> dTotals;                 // LmFrame::totals
    uint32_t hostCounters[LM_CNT_WORDS] = {0};
    bool countersValid = false;
    uint32_t lastDepth = 0;
    size_t lastLightCount = 0;

    // timing
    int timing = 0;                         // 0 off, 1 every kernel class, 2 closest-hit launches and the frame only (a quarter of the events)
    struct EvPair { hipEvent_t a, b; int cls; };
    std::vector<EvPair> evPool; size_t evUsed = 0;
    float classMs[6] = {0}; uint32_t classLaunches[6] = {0};
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

namespace lmr {

using R = lumen_mi_renderer;

// scene.cpp
void findEmissives(const R* r, Primitive& p);
int flatten(R* r);
void ensureWorldTris(R* r);
int syncScene(R* r, hipStream_t su);
int uploadResources(R* r);
int buildLights(R* r);
// frame.cpp
int ensureFrameBuffers(R* r);
int traceFrameAsync(R* r);
void cameraVectors(const float* right, const float* up, const float* fwd, float fovY, float aspect, float* U, float* V, float* Wv);
void motionMatrix(const float* prevCamWorld, float fovY, float aspect, float* M);
int syncAndCollect(R* r);
// launches the pending history passes of the last frame (R::owed) on `s`: mode 1 inside the next frame, 2 between frames
void launchOwedReuse(R* r, hipStream_t s, int mode);

}  // namespace lmr
