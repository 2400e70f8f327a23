"""Tile sharding of one frame over the GPUs of a node (new functionality; the reference is single-GPU).

One process per GPU.  The scene, BVH and light list are replicated; the image is cut into a cols x rows grid; every rank
renders a *window* = its tile grown by a halo of 60 px (clamped to the image), because ReSTIR's two spatial passes read
neighbours within +-30 px each (ReSTIRData.h:46-56).  RNG streams, Halton indices and light-bag tiles are functions of the
GLOBAL pixel coordinate, so a tile pixel whose whole 60-px neighbourhood lies inside the window gets exactly the
single-GPU value for a first frame; across blended frames the temporal history of the outer halo ring is window-local
(DESIGN.md §Multi-GPU states the seam behaviour).  The only collective is one gather of the final fp32 radiance tiles
(RCCL over xGMI when the backend is nccl; gloo on CPU in the tests); for path depths with temporal history an additional
point-to-point exchange of the halo rings' reservoirs after every frame keeps the seams exact (halo_plan, HistoryExchange below).
"""
import math

HALO = 60


def _plan(rank, n, width, height):
    from . import group
    return group.plan(width, height, n, rank)


def grid_for(n, width, height):
    """cols x rows with cols*rows == n whose LARGEST rank window (tile + halo, clipped to the image) is smallest: the frame time
    of the slowest rank is what the gather waits for.  One implementation: csrc/group.cpp (lumen_mi_group_plan)."""
    p = _plan(0, n, width, height)
    return p["cols"], p["rows"]


def tile_rect(rank, n, width, height):
    return _plan(rank, n, width, height)["tile"]


def window_rect(tile, width, height, halo=HALO):
    x0, y0, x1, y1 = tile
    return max(0, x0 - halo), max(0, y0 - halo), min(width, x1 + halo), min(height, y1 + halo)


def max_tile_shape(n, width, height):
    w, h = _plan(0, n, width, height)["max_tile"]
    return h, w


class TileGather:
    """The one collective of the tiled path: every rank's finished radiance tile to ``dst`` (RCCL gather over xGMI with the nccl backend).
    All buffers are allocated once: a send tile of the common (maximal) tile shape, the receive tiles and the assembled image on ``dst``;
    a frame costs one device copy into the send tile, one gather and ``world`` device copies into the image — no per-frame allocation."""

    def __init__(self, rank, world, width, height, dtype, device, dst=0):
        import torch
        self.rank, self.world, self.width, self.height, self.dst = rank, world, width, height, dst
        mh, mw = max_tile_shape(world, width, height)
        self.tile = tile_rect(rank, world, width, height)
        self.send = torch.zeros((mh, mw, 4), dtype=dtype, device=device)          # the padding is never read back
        self.parts = [torch.empty_like(self.send) for _ in range(world)] if (rank == dst and world > 1) else None
        self.image = torch.empty((height, width, 4), dtype=dtype, device=device) if rank == dst else None
        self.rects = [tile_rect(r, world, width, height) for r in range(world)]

    def run(self, local_tile, dist):
        """local_tile: tensor [th, tw, 4] (this rank's tile, halo removed).  Returns the assembled image on ``dst``, None elsewhere."""
        x0, y0, x1, y1 = self.tile
        if self.world == 1:
            self.image[y0:y1, x0:x1] = local_tile
            return self.image
        self.send[: y1 - y0, : x1 - x0] = local_tile
        dist.gather(self.send, self.parts, dst=self.dst)
        if self.rank != self.dst:
            return None
        for r, (rx0, ry0, rx1, ry1) in enumerate(self.rects):
            self.image[ry0:ry1, rx0:rx1] = self.parts[r][: ry1 - ry0, : rx1 - rx0]
        return self.image


    def run_renderer(self, renderer, dist):
        """The per-frame form: the rank's tile goes from the renderer's merged radiance straight into the send buffer, and on ``dst`` the gathered tiles into the
        assembled image, by the renderer library's own pitched-copy kernel on the renderer's stream (LumenRendererMI.CopyRadianceRectToDevice / CopyRectDevice) — no
        window-sized intermediate and no framework kernel in the frame path.  Returns the assembled image on ``dst``, None elsewhere.
        PRECONDITION: the renderer's stream is torch's current stream (``renderer.set_stream(torch.cuda.current_stream().cuda_stream)``): ``dist.gather`` orders itself
        against torch's current stream only, so a renderer on a stream of its own would have a half-written send tile gathered.  RGBA32F only (16 bytes per pixel).
        The native tile group (lumenrenderer_amd.group.TileGroup, csrc/group.cpp) has neither restriction and overlaps the gather with the next frame."""
        import torch
        assert self.send.dtype == torch.float32, "run_renderer moves RGBA32F pixels"
        x0, y0, x1, y1 = self.tile
        if self.world == 1:
            renderer.CopyRadianceRectToDevice(self.tile, self.image.data_ptr() + (y0 * self.width + x0) * 16, self.width)
            return self.image
        pitch = self.send.shape[1]
        renderer.CopyRadianceRectToDevice(self.tile, self.send.data_ptr(), pitch)
        dist.gather(self.send, self.parts, dst=self.dst)
        if self.rank != self.dst:
            return None
        for r, (rx0, ry0, rx1, ry1) in enumerate(self.rects):
            renderer.CopyRectDevice(self.image.data_ptr() + (ry0 * self.width + rx0) * 16, self.width, self.parts[r].data_ptr(), pitch, rx1 - rx0, ry1 - ry0)
        return self.image


_GATHERS = {}


def gather_tiles(local_tile, rank, world, width, height, dist, dst=0):
    """Convenience form of TileGather.run with the buffers cached per (shape, dtype, device)."""
    key = (rank, world, width, height, local_tile.dtype, str(local_tile.device), dst)
    g = _GATHERS.get(key)
    if g is None:
        g = _GATHERS[key] = TileGather(rank, world, width, height, local_tile.dtype, local_tile.device, dst)
    return g.run(local_tile, dist)


def gather_from_renderer(renderer, rank, world, width, height, dist, device, dst=0):
    """TileGather.run_renderer with the buffers cached per (rank, world, size, device)."""
    import torch
    key = (rank, world, width, height, torch.float32, str(device), dst, "renderer")
    g = _GATHERS.get(key)
    if g is None:
        g = _GATHERS[key] = TileGather(rank, world, width, height, torch.float32, device, dst)
    return g.run_renderer(renderer, dist)


# ---------------------------------------------------------------------------------------------------------------------
# temporal history across seams (option B for the history only)
# ---------------------------------------------------------------------------------------------------------------------
HISTORY_FLOATS = 20        # per pixel: the 64-byte reservoir record + the contribution plane (include/lumen_mi.h lumen_mi_export_history)


def halo_plan(rank, n, width, height):
    """What rank exchanges after every frame so that temporal reuse in its halo ring sees the owners' reservoirs: a list of
    (peer, send_rect, recv_rect) in global pixels — send = my tile inside the peer's window, recv = the peer's tile inside my
    window (either may be None).  The halo ring of a window is the disjoint union of the recv rectangles.  One implementation:
    csrc/group.cpp (lumen_mi_group_seams)."""
    from . import group
    return group.seams(width, height, n, rank)


def history_needed(depth):
    """With an even number of executed waves per frame the reference's buffer-swap quirk leaves temporal reuse without history
    (DESIGN.md §7), so there is nothing to exchange; the number of executed waves is the path depth unless the rays run out."""
    return depth % 2 == 1


def exchange_buffers(plan, send_buffers, recv_buffers, dist):
    """One grouped point-to-point exchange (RCCL send / recv over xGMI with the nccl backend; gloo in the CPU tests).
    send_buffers / recv_buffers: peer -> contiguous tensor.  Returns when the receives are complete on the current stream."""
    ops = []
    for peer, send, recv in plan:
        if recv is not None:
            ops.append(dist.P2POp(dist.irecv, recv_buffers[peer], peer))
        if send is not None:
            ops.append(dist.P2POp(dist.isend, send_buffers[peer], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class HistoryExchange:
    """Per-rank state of the seam exchange: the plan and its device staging buffers (allocated once)."""

    def __init__(self, renderer, rank, world, width, height, device):
        import torch
        self.r, self.plan = renderer, halo_plan(rank, world, width, height)
        area = lambda q: (q[2] - q[0]) * (q[3] - q[1])
        self.waves = torch.zeros(1, dtype=torch.int32, device=device)
        self.send = {p: torch.empty(area(s) * HISTORY_FLOATS, dtype=torch.float32, device=device) for p, s, _ in self.plan if s}
        self.recv = {p: torch.empty(area(q) * HISTORY_FLOATS, dtype=torch.float32, device=device) for p, _, q in self.plan if q}

    def pack(self):
        for peer, send, _ in self.plan:
            if send:
                self.r.ExportHistory(send, self.send[peer].data_ptr())

    def unpack(self):
        for peer, _, recv in self.plan:
            if recv:
                self.r.ImportHistory(recv, self.recv[peer].data_ptr())

    def run(self, dist):
        """Call after every TraceFrame(Async): everything is stream-ordered behind the frame's merge and ahead of the next frame's
        temporal pass (the renderer's stream must be torch's current stream: LumenRendererMI.set_stream)."""
        self.r.ExportWaveCount(self.waves.data_ptr())          # ranks whose rays ran out at different depths agree on the swap chain
        dist.all_reduce(self.waves, op=dist.ReduceOp.MAX)
        self.r.ImportWaveCount(self.waves.data_ptr())
        self.pack()                                            # ... before the buffer the next frame calls "previous" is picked
        exchange_buffers(self.plan, self.send, self.recv, dist)
        self.unpack()
