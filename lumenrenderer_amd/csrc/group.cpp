// group.cpp — the tiled multi-GPU frame behind the C ABI (include/lumen_mi.h "tile groups"): one process per GPU, N processes render one
// image.  New functionality — the reference is single-GPU; north_star: "frames shard by tile across the 8 GPUs of one node
// with RCCL gather over xGMI of the final radiance buffer", with a C++ host.  Everything a rank does per frame is here, in C++:
//   plan      cols x rows grid whose worst window (tile + 60-px halo, clipped) is smallest; tile, window, the common send-tile shape, the seam plan
//   frame     TraceFrame of the window; at odd path depths (the only ones whose temporal history is ever read) the ranks agree on the
//             executed wave count (all-reduce MAX) and exchange the halo rings' reservoirs in ONE grouped send / recv (80 B per pixel)
//   gather    the rank's tile leaves the merged radiance through lm_k_copy_rect into one of TWO send tiles, and travels on a second stream: grouped
//             ncclSend / ncclRecv to rank 0, which places every tile in the assembled frame with the same kernel.  Double buffering + the second
//             stream let gather(i) overlap the rendering of frame i + 1.
// Transport: RCCL, resolved at group creation with dlopen("librccl.so.1") — the library itself keeps no link-time dependency on RCCL, so a single-GPU
// host needs none, and inside a PyTorch process the copy PyTorch has already loaded is the one that is used.  A HOST transport can be injected instead
// (lumen_mi_transport: callbacks on host buffers): the same plan, buffers, kernels and call order with the bytes staged through pinned memory — what the
// suite uses to run 2 / 4 / 8 ranks on the one GPU of a development box, where RCCL refuses several ranks per device.
#include "renderer_state.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace lmr;

namespace {

constexpr uint32_t HALO = 60;                 // two spatial reuse passes x 30 px (ReSTIRData.h:46-56)
constexpr uint32_t HISTORY_BYTES = 80;        // per pixel: 64-byte reservoir record + contribution plane (lumen_mi_export_history)

struct Rect { uint32_t x0, y0, x1, y1; bool empty() const { return x0 >= x1 || y0 >= y1; } uint32_t w() const { return x1 - x0; } uint32_t h() const { return y1 - y0; } size_t area() const { return (size_t)w() * h(); } };

Rect tileOf(uint32_t cols, uint32_t rows, uint32_t rank, uint32_t W, uint32_t H)
{
    const uint32_t cx = rank % cols, cy = rank / cols;
    return Rect{(uint32_t)(((uint64_t)W * cx) / cols), (uint32_t)(((uint64_t)H * cy) / rows), (uint32_t)(((uint64_t)W * (cx + 1)) / cols), (uint32_t)(((uint64_t)H * (cy + 1)) / rows)};
}
Rect windowOf(const Rect& t, uint32_t W, uint32_t H)
{
    return Rect{t.x0 > HALO ? t.x0 - HALO : 0u, t.y0 > HALO ? t.y0 - HALO : 0u, std::min(W, t.x1 + HALO), std::min(H, t.y1 + HALO)};
}
Rect intersect(const Rect& a, const Rect& b)
{
    Rect r{std::max(a.x0, b.x0), std::max(a.y0, b.y0), std::min(a.x1, b.x1), std::min(a.y1, b.y1)};
    if (r.empty()) r = Rect{0, 0, 0, 0};
    return r;
}
// the grid whose LARGEST rank window is smallest: the frame time of the slowest rank is what the gather waits for (first such grid in order of cols)
void gridFor(uint32_t n, uint32_t W, uint32_t H, uint32_t& colsOut, uint32_t& rowsOut)
{
    uint64_t best = ~0ull; colsOut = n; rowsOut = 1;
    for (uint32_t cols = 1; cols <= n; cols++) {
        if (n % cols) continue;
        const uint32_t rows = n / cols;
        uint64_t worst = 0;
        for (uint32_t r = 0; r < n; r++) { const Rect w = windowOf(tileOf(cols, rows, r, W, H), W, H); worst = std::max<uint64_t>(worst, w.area()); }
        if (worst < best) { best = worst; colsOut = cols; rowsOut = rows; }
    }
}

// ---- RCCL, resolved at run time ---------------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl* rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("LUMEN_MI_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
        for (const char* n : names) { if (n && *n && (R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break; }
        if (!R.lib) { R.error = std::string("RCCL is not loadable (") + (dlerror() ? dlerror() : "dlopen failed") + "); set LUMEN_MI_RCCL_LIBRARY or inject a host transport"; return; }
        auto sym = [&](const char* s) { void* p = dlsym(R.lib, s); if (!p && R.error.empty()) R.error = std::string("librccl lacks ") + s; return p; };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId"); R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy"); R.CommAbort = (decltype(R.CommAbort))sym("ncclCommAbort");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart"); R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.Send = (decltype(R.Send))sym("ncclSend"); R.Recv = (decltype(R.Recv))sym("ncclRecv"); R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    return &R;
}
#define LM_NCCL(expr)                                                                                                       \
    do {                                                                                                                    \
        ncclResult_t e_ = (expr);                                                                                           \
        if (e_ != ncclSuccess) return fail(LUMEN_MI_ERR_DEVICE, std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(e_) : "RCCL error") + " (rank " + std::to_string(g->rank) + " of " + std::to_string(g->world) + ")"); \
    } while (0)

struct Seam { uint32_t peer; Rect send, recv; void* dSend = nullptr; void* dRecv = nullptr; void* hSend = nullptr; void* hRecv = nullptr; };

}  // namespace

struct lumen_mi_group {
    lumen_mi_renderer* r = nullptr;
    uint32_t rank = 0, world = 1, W = 0, H = 0, cols = 1, rows = 1, maxTw = 0, maxTh = 0;
    Rect tile{}, window{};
    std::vector<Rect> tiles;
    std::vector<Seam> seams;
    bool rcclMode = false, hostMode = false;
    lumen_mi_transport host{};
    ncclComm_t commFrame = nullptr, commGather = nullptr;      // one communicator per stream: seam traffic and the gather never wait for each other
    hipStream_t gatherStream = nullptr;
    hipEvent_t tileReady[2] = {nullptr, nullptr}, gatherDone[2] = {nullptr, nullptr};
    float4* send[2] = {nullptr, nullptr};        // [maxTh][maxTw] RGBA32F
    float4* parts[2] = {nullptr, nullptr};       // rank 0: world tiles of the common shape per parity
    float4* image[2] = {nullptr, nullptr};       // rank 0: the assembled frame per parity
    void* hSend = nullptr; void* hParts = nullptr;           // host transport: pinned staging
    int32_t* dWaves = nullptr; int32_t* hWaves = nullptr;
    uint32_t parity = 0; int lastParity = -1; uint64_t gathers = 0;
    bool gatherPending[2] = {false, false};
    float gatherMs = 0.f; uint32_t gatherTimed = 0; hipEvent_t gatherStart[2] = {nullptr, nullptr};
};

extern "C" {

int lumen_mi_group_plan(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, lumen_mi_tile_plan* out)
{
    if (!out || !width || !height || !world || rank >= world) return fail(LUMEN_MI_ERR_INVALID, "lumen_mi_group_plan: need an image, world >= 1 and rank < world");
    if ((uint64_t)world > (uint64_t)width * height) return fail(LUMEN_MI_ERR_INVALID, "more ranks than pixels");
    uint32_t cols, rows;
    gridFor(world, width, height, cols, rows);
    const Rect t = tileOf(cols, rows, rank, width, height), w = windowOf(t, width, height);
    if (t.empty()) return fail(LUMEN_MI_ERR_INVALID, "the image is too small for this many ranks: a tile would be empty");
    out->cols = cols; out->rows = rows; out->halo = HALO;
    out->tile[0] = t.x0; out->tile[1] = t.y0; out->tile[2] = t.x1; out->tile[3] = t.y1;
    out->window[0] = w.x0; out->window[1] = w.y0; out->window[2] = w.x1; out->window[3] = w.y1;
    out->max_tile_w = out->max_tile_h = 0;
    for (uint32_t r = 0; r < world; r++) { const Rect q = tileOf(cols, rows, r, width, height); out->max_tile_w = std::max(out->max_tile_w, q.w()); out->max_tile_h = std::max(out->max_tile_h, q.h()); }
    return 0;
}

int lumen_mi_group_seams(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, lumen_mi_seam* out, uint32_t capacity, uint32_t* count)
{
    if (!count || !width || !height || !world || rank >= world) return fail(LUMEN_MI_ERR_INVALID, "lumen_mi_group_seams: need an image, world >= 1, rank < world and a count pointer");
    uint32_t cols, rows;
    gridFor(world, width, height, cols, rows);
    const Rect mine = tileOf(cols, rows, rank, width, height), myWindow = windowOf(mine, width, height);
    uint32_t n = 0;
    for (uint32_t peer = 0; peer < world; peer++) {
        if (peer == rank) continue;
        const Rect theirs = tileOf(cols, rows, peer, width, height);
        const Rect s = intersect(mine, windowOf(theirs, width, height)), q = intersect(theirs, myWindow);
        if (s.empty() && q.empty()) continue;
        if (out && n < capacity) {
            out[n].peer = peer;
            out[n].send[0] = s.x0; out[n].send[1] = s.y0; out[n].send[2] = s.x1; out[n].send[3] = s.y1;
            out[n].recv[0] = q.x0; out[n].recv[1] = q.y0; out[n].recv[2] = q.x1; out[n].recv[3] = q.y1;
        }
        n++;
    }
    *count = n;
    if (out && n > capacity) return fail(LUMEN_MI_ERR_INVALID, "seam capacity too small");
    return 0;
}

int lumen_mi_group_unique_id(uint8_t id[LUMEN_MI_GROUP_ID_BYTES])
{
    if (!id) return fail(LUMEN_MI_ERR_INVALID, "NULL id");
    Rccl* R = rccl();
    if (!R->error.empty()) return fail(LUMEN_MI_ERR_STATE, R->error);
    static_assert(LUMEN_MI_GROUP_ID_BYTES == 2 * sizeof(ncclUniqueId), "two communicator ids");
    for (int k = 0; k < 2; k++) {
        ncclUniqueId u;
        const ncclResult_t e = R->GetUniqueId(&u);
        if (e != ncclSuccess) return fail(LUMEN_MI_ERR_DEVICE, std::string("ncclGetUniqueId: ") + R->GetErrorString(e));
        memcpy(id + k * sizeof u, &u, sizeof u);
    }
    return 0;
}

static void freeGroup(lumen_mi_group* g)
{
    if (!g) return;
    if (g->r) (void)hipSetDevice(g->r->device);
    if (g->gatherStream) (void)hipStreamSynchronize(g->gatherStream);
    Rccl* R = rccl();
    if (g->commFrame && R->CommDestroy) R->CommDestroy(g->commFrame);
    if (g->commGather && R->CommDestroy) R->CommDestroy(g->commGather);
    for (int k = 0; k < 2; k++) {
        if (g->send[k]) (void)hipFree(g->send[k]);
        if (g->parts[k]) (void)hipFree(g->parts[k]);
        if (g->image[k]) (void)hipFree(g->image[k]);
        if (g->tileReady[k]) (void)hipEventDestroy(g->tileReady[k]);
        if (g->gatherDone[k]) (void)hipEventDestroy(g->gatherDone[k]);
        if (g->gatherStart[k]) (void)hipEventDestroy(g->gatherStart[k]);
    }
    for (Seam& s : g->seams) {
        if (s.dSend) (void)hipFree(s.dSend);
        if (s.dRecv) (void)hipFree(s.dRecv);
        if (s.hSend) (void)hipHostFree(s.hSend);
        if (s.hRecv) (void)hipHostFree(s.hRecv);
    }
    if (g->hSend) (void)hipHostFree(g->hSend);
    if (g->hParts) (void)hipHostFree(g->hParts);
    if (g->dWaves) (void)hipFree(g->dWaves);
    if (g->hWaves) (void)hipHostFree(g->hWaves);
    if (g->gatherStream) (void)hipStreamDestroy(g->gatherStream);
    delete g;
}

int lumen_mi_group_create(lumen_mi_renderer* r, uint32_t rank, uint32_t world, const uint8_t* id, const lumen_mi_transport* transport, lumen_mi_group** out)
{
    if (!r || !out || !world || rank >= world) return fail(LUMEN_MI_ERR_INVALID, "lumen_mi_group_create: need a renderer, world >= 1 and rank < world");
    if (transport && (!transport->exchange || !transport->allreduce_max_i32)) return fail(LUMEN_MI_ERR_INVALID, "a host transport needs both callbacks");
    if (!transport && world > 1 && !id) return fail(LUMEN_MI_ERR_INVALID, "RCCL transport: every rank needs the id rank 0 obtained from lumen_mi_group_unique_id");
    uint32_t W = 0, H = 0;
    if (int e = lumen_mi_get_render_resolution(r, &W, &H)) return e;
    lumen_mi_tile_plan plan;
    if (int e = lumen_mi_group_plan(W, H, world, rank, &plan)) return e;
    std::unique_ptr<lumen_mi_group, void (*)(lumen_mi_group*)> g(new lumen_mi_group, freeGroup);
    g->r = r; g->rank = rank; g->world = world; g->W = W; g->H = H; g->cols = plan.cols; g->rows = plan.rows; g->maxTw = plan.max_tile_w; g->maxTh = plan.max_tile_h;
    g->tile = Rect{plan.tile[0], plan.tile[1], plan.tile[2], plan.tile[3]}; g->window = Rect{plan.window[0], plan.window[1], plan.window[2], plan.window[3]};
    for (uint32_t k = 0; k < world; k++) g->tiles.push_back(tileOf(plan.cols, plan.rows, k, W, H));
    if (transport) { g->hostMode = true; g->host = *transport; }
    else g->rcclMode = world > 1 || id != nullptr;           // world == 1 without an id: no transport at all (the frame is the tile)
    if (world > 1) {
        if (int e = lumen_mi_set_window(r, g->window.x0, g->window.y0, g->window.x1, g->window.y1)) return e;
        if (int e = lumen_mi_set_tile(r, g->tile.x0, g->tile.y0, g->tile.x1, g->tile.y1)) return e;
    }
    int device;
    { ApiLock lk(r); if (!r->initialised) return fail(LUMEN_MI_ERR_STATE, "not initialised"); device = r->device; }
    if (hipSetDevice(device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    LM_HIP(hipStreamCreateWithFlags(&g->gatherStream, hipStreamNonBlocking));
    const size_t tileBytes = (size_t)g->maxTw * g->maxTh * sizeof(float4);
    for (int k = 0; k < 2; k++) {
        LM_HIP(hipEventCreateWithFlags(&g->tileReady[k], hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags(&g->gatherDone[k], hipEventDefault));
        LM_HIP(hipEventCreateWithFlags(&g->gatherStart[k], hipEventDefault));
        LM_HIP(hipMalloc((void**)&g->send[k], tileBytes));
        LM_HIP(hipMemset(g->send[k], 0, tileBytes));                                  // the padding of a smaller tile travels but is never read
        if (rank == 0) {
            if (world > 1) LM_HIP(hipMalloc((void**)&g->parts[k], tileBytes * world));
            LM_HIP(hipMalloc((void**)&g->image[k], (size_t)W * H * sizeof(float4)));
        }
    }
    LM_HIP(hipMalloc((void**)&g->dWaves, sizeof(int32_t)));
    if (g->hostMode) {
        LM_HIP(hipHostMalloc(&g->hSend, tileBytes, hipHostMallocDefault));
        if (rank == 0 && world > 1) LM_HIP(hipHostMalloc(&g->hParts, tileBytes * world, hipHostMallocDefault));
        LM_HIP(hipHostMalloc((void**)&g->hWaves, sizeof(int32_t), hipHostMallocDefault));
    }
    // seam plan and its staging buffers (used at odd path depths only; small: the halo ring)
    uint32_t ns = 0;
    if (int e = lumen_mi_group_seams(W, H, world, rank, nullptr, 0, &ns)) return e;
    std::vector<lumen_mi_seam> raw(ns);
    if (ns) { if (int e = lumen_mi_group_seams(W, H, world, rank, raw.data(), ns, &ns)) return e; }
    for (const lumen_mi_seam& q : raw) {
        Seam s; s.peer = q.peer; s.send = Rect{q.send[0], q.send[1], q.send[2], q.send[3]}; s.recv = Rect{q.recv[0], q.recv[1], q.recv[2], q.recv[3]};
        g->seams.push_back(s);
        Seam& t = g->seams.back();                       // (pushed first: freeGroup releases whatever has been allocated when a later allocation fails)
        if (!t.send.empty()) { LM_HIP(hipMalloc(&t.dSend, t.send.area() * HISTORY_BYTES)); if (g->hostMode) LM_HIP(hipHostMalloc(&t.hSend, t.send.area() * HISTORY_BYTES, hipHostMallocDefault)); }
        if (!t.recv.empty()) { LM_HIP(hipMalloc(&t.dRecv, t.recv.area() * HISTORY_BYTES)); if (g->hostMode) LM_HIP(hipHostMalloc(&t.hRecv, t.recv.area() * HISTORY_BYTES, hipHostMallocDefault)); }
    }
    if (g->rcclMode) {
        Rccl* R = rccl();
        if (!R->error.empty()) return fail(LUMEN_MI_ERR_STATE, R->error);
        uint8_t self[LUMEN_MI_GROUP_ID_BYTES];
        if (!id) { if (int e = lumen_mi_group_unique_id(self)) return e; id = self; }
        ncclUniqueId u0, u1;
        memcpy(&u0, id, sizeof u0); memcpy(&u1, id + sizeof u0, sizeof u1);
        LM_NCCL(R->CommInitRank(&g->commFrame, (int)world, u0, (int)rank));
        LM_NCCL(R->CommInitRank(&g->commGather, (int)world, u1, (int)rank));
    }
    *out = g.release();
    return 0;
}

int lumen_mi_group_destroy(lumen_mi_group* g)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    if (g->world > 1) { (void)lumen_mi_set_tile(g->r, 0, 0, 0, 0); (void)lumen_mi_set_window(g->r, 0, 0, 0, 0); }
    freeGroup(g);
    return 0;
}

int lumen_mi_group_get_plan(lumen_mi_group* g, lumen_mi_tile_plan* out)
{
    if (!g || !out) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    return lumen_mi_group_plan(g->W, g->H, g->world, g->rank, out);
}

// the seam step of one frame: wave-count agreement, then ONE grouped exchange of the halo rings' reservoirs; everything on the renderer's stream, behind the
// frame's last kernel and ahead of the next frame's temporal pass
static int exchangeHistory(lumen_mi_group* g)
{
    lumen_mi_renderer* r = g->r;
    hipStream_t st;
    { ApiLock lk(r); st = r->stream; }
    if (int e = lumen_mi_export_wave_count(r, g->dWaves)) return e;
    if (g->rcclMode) {
        Rccl* R = rccl();
        LM_NCCL(R->AllReduce(g->dWaves, g->dWaves, 1, ncclInt32, ncclMax, g->commFrame, st));
    } else {
        LM_HIP(hipMemcpyAsync(g->hWaves, g->dWaves, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        if (g->host.allreduce_max_i32(g->host.user, g->hWaves)) return fail(LUMEN_MI_ERR_DEVICE, "host transport: allreduce_max_i32 failed (rank " + std::to_string(g->rank) + ")");
        LM_HIP(hipMemcpyAsync(g->dWaves, g->hWaves, sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    if (int e = lumen_mi_import_wave_count(r, g->dWaves)) return e;
    for (Seam& s : g->seams)
        if (!s.send.empty()) { if (int e = lumen_mi_export_history(r, s.send.x0, s.send.y0, s.send.x1, s.send.y1, s.dSend)) return e; }
    if (g->rcclMode) {
        Rccl* R = rccl();
        LM_NCCL(R->GroupStart());
        for (Seam& s : g->seams) {
            if (!s.recv.empty()) LM_NCCL(R->Recv(s.dRecv, s.recv.area() * HISTORY_BYTES, ncclUint8, (int)s.peer, g->commFrame, st));
            if (!s.send.empty()) LM_NCCL(R->Send(s.dSend, s.send.area() * HISTORY_BYTES, ncclUint8, (int)s.peer, g->commFrame, st));
        }
        LM_NCCL(R->GroupEnd());
    } else {
        std::vector<lumen_mi_transport_op> ops;
        for (Seam& s : g->seams) {
            if (!s.recv.empty()) ops.push_back(lumen_mi_transport_op{s.peer, 0, s.hRecv, s.recv.area() * HISTORY_BYTES});
            if (!s.send.empty()) { LM_HIP(hipMemcpyAsync(s.hSend, s.dSend, s.send.area() * HISTORY_BYTES, hipMemcpyDeviceToHost, st)); ops.push_back(lumen_mi_transport_op{s.peer, 1, s.hSend, s.send.area() * HISTORY_BYTES}); }
        }
        LM_HIP(hipStreamSynchronize(st));
        if (!ops.empty() && g->host.exchange(g->host.user, (uint32_t)ops.size(), ops.data())) return fail(LUMEN_MI_ERR_DEVICE, "host transport: exchange failed (rank " + std::to_string(g->rank) + ")");
        for (Seam& s : g->seams)
            if (!s.recv.empty()) LM_HIP(hipMemcpyAsync(s.dRecv, s.hRecv, s.recv.area() * HISTORY_BYTES, hipMemcpyHostToDevice, st));
    }
    for (Seam& s : g->seams)
        if (!s.recv.empty()) { if (int e = lumen_mi_import_history(r, s.recv.x0, s.recv.y0, s.recv.x1, s.recv.y1, s.dRecv)) return e; }
    return 0;
}

int lumen_mi_group_trace_frame(lumen_mi_group* g)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    lumen_mi_renderer* r = g->r;
    const int e = lumen_mi_trace_frame_async(r);
    if (e) return e;
    uint32_t depth;
    { std::lock_guard<std::mutex> sl(r->settingsMutex); depth = r->pending.depth; }
    // an even number of executed waves leaves temporal reuse without history (the reference's swap quirk, WaveFrontRenderer.cpp:827): nothing to exchange
    if (g->world > 1 && (depth & 1u)) return exchangeHistory(g);
    return 0;
}

int lumen_mi_group_gather(lumen_mi_group* g)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    lumen_mi_renderer* r = g->r;
    hipStream_t st; int device; const LmKernelTable* K;
    { ApiLock lk(r); st = r->stream; device = r->device; K = r->K; }
    if (hipSetDevice(device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t p = g->parity;
    const Rect& t = g->tile;
    // the send tile of this parity is free once the gather that last used it has finished
    if (g->gatherPending[p]) {
        LM_HIP(hipStreamWaitEvent(st, g->gatherDone[p], 0));
        float ms = 0.f;           // that gather's device time, if it is over already (two frames back: it normally is); never waited for
        if (g->world > 1 && hipEventQuery(g->gatherDone[p]) == hipSuccess && hipEventElapsedTime(&ms, g->gatherStart[p], g->gatherDone[p]) == hipSuccess) { g->gatherMs += ms; g->gatherTimed++; }
        (void)hipGetLastError();  // (hipErrorNotReady of the query is not an error of this call)
    }
    if (g->world == 1) {          // the frame is the tile: straight into the assembled image
        if (int e = lumen_mi_copy_radiance_rect_device(r, t.x0, t.y0, t.x1, t.y1, g->image[p] + (size_t)t.y0 * g->W + t.x0, g->W)) return e;
        LM_HIP(hipEventRecord(g->gatherDone[p], st));
        g->gatherPending[p] = true; g->lastParity = (int)p; g->parity ^= 1u; g->gathers++;
        return 0;
    }
    if (int e = lumen_mi_copy_radiance_rect_device(r, t.x0, t.y0, t.x1, t.y1, g->send[p], g->maxTw)) return e;
    LM_HIP(hipEventRecord(g->tileReady[p], st));
    hipStream_t gs = g->gatherStream;
    LM_HIP(hipStreamWaitEvent(gs, g->tileReady[p], 0));
    LM_HIP(hipEventRecord(g->gatherStart[p], gs));
    const size_t tileFloats = (size_t)g->maxTw * g->maxTh * 4, tileBytes = tileFloats * sizeof(float);
    if (g->rcclMode) {
        Rccl* R = rccl();
        LM_NCCL(R->GroupStart());
        if (g->rank == 0) { for (uint32_t k = 1; k < g->world; k++) LM_NCCL(R->Recv(g->parts[p] + (size_t)k * g->maxTw * g->maxTh, tileFloats, ncclFloat, (int)k, g->commGather, gs)); }
        else LM_NCCL(R->Send(g->send[p], tileFloats, ncclFloat, 0, g->commGather, gs));
        LM_NCCL(R->GroupEnd());
    } else {
        std::vector<lumen_mi_transport_op> ops;
        if (g->rank == 0) { for (uint32_t k = 1; k < g->world; k++) ops.push_back(lumen_mi_transport_op{k, 0, (char*)g->hParts + (size_t)k * tileBytes, tileBytes}); }
        else { LM_HIP(hipMemcpyAsync(g->hSend, g->send[p], tileBytes, hipMemcpyDeviceToHost, gs)); ops.push_back(lumen_mi_transport_op{0u, 1, g->hSend, tileBytes}); }
        LM_HIP(hipStreamSynchronize(gs));
        if (g->host.exchange(g->host.user, (uint32_t)ops.size(), ops.data())) return fail(LUMEN_MI_ERR_DEVICE, "host transport: exchange failed (rank " + std::to_string(g->rank) + ")");
        if (g->rank == 0) LM_HIP(hipMemcpyAsync(g->parts[p] + (size_t)g->maxTw * g->maxTh, (char*)g->hParts + tileBytes, tileBytes * (g->world - 1), hipMemcpyHostToDevice, gs));
    }
    if (g->rank == 0) {
        for (uint32_t k = 0; k < g->world; k++) {
            const Rect& q = g->tiles[k];
            const float4* src = k == 0 ? g->send[p] : g->parts[p] + (size_t)k * g->maxTw * g->maxTh;
            K->copy_rect(gs, r->gridFor(q.w() * q.h(), 8), g->image[p] + (size_t)q.y0 * g->W + q.x0, g->W, src, g->maxTw, q.w(), q.h());
        }
        LM_HIP(hipGetLastError());
    }
    LM_HIP(hipEventRecord(g->gatherDone[p], gs));
    g->gatherPending[p] = true; g->lastParity = (int)p; g->parity ^= 1u; g->gathers++;
    return 0;
}

int lumen_mi_group_synchronize(lumen_mi_group* g)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    if (int e = lumen_mi_synchronize(g->r)) return e;
    if (hipSetDevice(g->r->device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    LM_HIP(hipStreamSynchronize(g->gatherStream));
    return 0;
}

int lumen_mi_group_frame_device(lumen_mi_group* g, void** device_rgba32f)
{
    if (!g || !device_rgba32f) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    if (g->rank != 0) return fail(LUMEN_MI_ERR_STATE, "only rank 0 holds the assembled frame");
    if (g->lastParity < 0) return fail(LUMEN_MI_ERR_STATE, "nothing has been gathered yet");
    if (int e = lumen_mi_group_synchronize(g)) return e;
    *device_rgba32f = g->image[g->lastParity];
    return 0;
}

int lumen_mi_group_get_frame(lumen_mi_group* g, float* rgba32f, size_t capacity_bytes)
{
    if (!g || !rgba32f) return fail(LUMEN_MI_ERR_INVALID, "NULL argument");
    const size_t bytes = (size_t)g->W * g->H * sizeof(float4);
    if (capacity_bytes < bytes) return fail(LUMEN_MI_ERR_INVALID, "buffer too small for the frame");
    void* dev = nullptr;
    if (int e = lumen_mi_group_frame_device(g, &dev)) return e;
    LM_HIP(hipMemcpy(rgba32f, dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// Before a timed run: one 1-element all-reduce on each communicator and one full-size gather of the (still empty) send tiles, synchronised, with the time they took.
// A rank that cannot reach its peers fails HERE, with its rank in the message, instead of hanging the first frame.
int lumen_mi_group_self_test(lumen_mi_group* g, float* milliseconds)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    lumen_mi_renderer* r = g->r;
    hipStream_t st; int device;
    { ApiLock lk(r); st = r->stream; device = r->device; }
    if (hipSetDevice(device) != hipSuccess) return fail(LUMEN_MI_ERR_DEVICE, "hipSetDevice failed");
    const auto t0 = std::chrono::steady_clock::now();
    LM_HIP(hipMemsetAsync(g->dWaves, 0, sizeof(int32_t), st));
    int32_t sum = -1;
    if (g->rcclMode) {
        Rccl* R = rccl();
        int32_t one = 1;
        LM_HIP(hipMemcpyAsync(g->dWaves, &one, sizeof one, hipMemcpyHostToDevice, st));
        LM_NCCL(R->AllReduce(g->dWaves, g->dWaves, 1, ncclInt32, ncclSum, g->commFrame, st));
        LM_HIP(hipStreamSynchronize(st));
        LM_HIP(hipMemcpy(&sum, g->dWaves, sizeof sum, hipMemcpyDeviceToHost));
        if (sum != (int32_t)g->world) return fail(LUMEN_MI_ERR_DEVICE, "group self-test: all-reduce over " + std::to_string(g->world) + " ranks returned " + std::to_string(sum) + " on rank " + std::to_string(g->rank));
        LM_NCCL(R->AllReduce(g->dWaves, g->dWaves, 1, ncclInt32, ncclMax, g->commGather, g->gatherStream));
        LM_HIP(hipStreamSynchronize(g->gatherStream));
    } else if (g->hostMode) {
        int32_t v = (int32_t)g->rank;
        if (g->host.allreduce_max_i32(g->host.user, &v) || v != (int32_t)g->world - 1) return fail(LUMEN_MI_ERR_DEVICE, "group self-test: host all-reduce MAX of the ranks returned " + std::to_string(v) + " on rank " + std::to_string(g->rank));
    }
    if (g->world > 1) {
        // the gather of two frames' worth of (empty) tiles, both parities: allocations, events and the transport at full size
        for (int k = 0; k < 2; k++) {
            const uint32_t p = g->parity;
            hipStream_t gs = g->gatherStream;
            const size_t tileFloats = (size_t)g->maxTw * g->maxTh * 4, tileBytes = tileFloats * sizeof(float);
            if (g->rcclMode) {
                Rccl* R = rccl();
                LM_NCCL(R->GroupStart());
                if (g->rank == 0) { for (uint32_t q = 1; q < g->world; q++) LM_NCCL(R->Recv(g->parts[p] + (size_t)q * g->maxTw * g->maxTh, tileFloats, ncclFloat, (int)q, g->commGather, gs)); }
                else LM_NCCL(R->Send(g->send[p], tileFloats, ncclFloat, 0, g->commGather, gs));
                LM_NCCL(R->GroupEnd());
            } else {
                std::vector<lumen_mi_transport_op> ops;
                if (g->rank == 0) { for (uint32_t q = 1; q < g->world; q++) ops.push_back(lumen_mi_transport_op{q, 0, (char*)g->hParts + (size_t)q * tileBytes, tileBytes}); }
                else { LM_HIP(hipMemcpyAsync(g->hSend, g->send[p], tileBytes, hipMemcpyDeviceToHost, gs)); ops.push_back(lumen_mi_transport_op{0u, 1, g->hSend, tileBytes}); }
                LM_HIP(hipStreamSynchronize(gs));
                if (g->host.exchange(g->host.user, (uint32_t)ops.size(), ops.data())) return fail(LUMEN_MI_ERR_DEVICE, "group self-test: host exchange failed (rank " + std::to_string(g->rank) + ")");
            }
            LM_HIP(hipStreamSynchronize(gs));
            g->parity ^= 1u;
        }
    }
    if (milliseconds) *milliseconds = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int lumen_mi_group_get_stats(lumen_mi_group* g, uint64_t* gathers, float* mean_gather_ms)
{
    if (!g) return fail(LUMEN_MI_ERR_INVALID, "NULL group");
    if (gathers) *gathers = g->gathers;
    if (mean_gather_ms) *mean_gather_ms = g->gatherTimed ? g->gatherMs / (float)g->gatherTimed : 0.f;
    return 0;
}

}  // extern "C"
