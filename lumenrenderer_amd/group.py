"""Tile groups: N processes, one per GPU, render ONE image — ctypes face of include/lumen_mi.h "tile groups" (csrc/group.cpp: the tile / window / halo plan,
the seam exchange and the double-buffered gather live in C++ behind the C ABI; this file only moves the communicator id between the ranks and, for the
one-GPU rehearsals of the suite, offers a host transport made of ``torch.distributed`` point-to-point calls on CPU tensors)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check


def plan(width, height, world, rank):
    """The C++ tile plan of one rank (pure function, no GPU): dict with cols, rows, halo, tile, window, max_tile (w, h)."""
    lib = capi.load_library()
    p = capi.TilePlan()
    check(lib, lib.lumen_mi_group_plan(width, height, world, rank, C.byref(p)))
    return {"cols": p.cols, "rows": p.rows, "halo": p.halo, "tile": tuple(p.tile), "window": tuple(p.window), "max_tile": (p.max_tile_w, p.max_tile_h)}


def seams(width, height, world, rank):
    """[(peer, send_rect | None, recv_rect | None)] of the C++ seam plan (global pixels)."""
    lib = capi.load_library()
    n = C.c_uint32(0)
    check(lib, lib.lumen_mi_group_seams(width, height, world, rank, None, 0, C.byref(n)))
    arr = (capi.Seam * max(1, n.value))()
    check(lib, lib.lumen_mi_group_seams(width, height, world, rank, arr, n.value, C.byref(n)))
    rect = lambda q: tuple(q) if (q[2] > q[0] and q[3] > q[1]) else None
    return [(arr[i].peer, rect(arr[i].send), rect(arr[i].recv)) for i in range(n.value)]


def unique_id():
    """256 bytes from RCCL on the calling rank (rank 0), to be handed to every other rank."""
    lib = capi.load_library()
    buf = (C.c_uint8 * capi.GROUP_ID_BYTES)()
    check(lib, lib.lumen_mi_group_unique_id(buf))
    return bytes(buf)


class DistHostTransport:
    """lumen_mi_transport over a ``torch.distributed`` process group whose backend moves CPU tensors (gloo): the exchange is ONE batch_isend_irecv, the
    all-reduce is dist.all_reduce(MAX).  For rehearsing several ranks on one GPU (RCCL refuses that); on a multi-GPU node the group talks RCCL itself."""

    def __init__(self, dist):
        import torch
        self._dist, self._torch = dist, torch
        self.errors = []

        def exchange(user, n, ops):
            try:
                reqs, keep = [], []
                for i in range(n):
                    o = ops[i]
                    t = torch.frombuffer((C.c_uint8 * o.bytes).from_address(o.host), dtype=torch.uint8)
                    keep.append(t)
                    reqs.append(dist.P2POp(dist.isend if o.send else dist.irecv, t, int(o.peer)))
                if reqs:
                    for r in dist.batch_isend_irecv(reqs):
                        r.wait()
                return 0
            except Exception as ex:                       # an exception must not unwind through the C frames
                self.errors.append(repr(ex))
                return 1

        def allreduce(user, value):
            try:
                t = torch.tensor([value[0]], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                value[0] = int(t[0])
                return 0
            except Exception as ex:
                self.errors.append(repr(ex))
                return 1

        self._fns = (capi.EXCHANGE_FN(exchange), capi.ALLREDUCE_FN(allreduce))      # kept alive as long as the transport
        self.struct = capi.Transport(None, self._fns[0], self._fns[1])


class TileGroup:
    """One rank of a tile group around a LumenRendererMI that was initialised at the FULL image resolution."""

    def __init__(self, renderer, rank, world, group_id=None, transport=None):
        self.r, self.rank, self.world, self.lib = renderer, rank, world, renderer.lib
        self._transport = transport
        idbuf = (C.c_uint8 * capi.GROUP_ID_BYTES).from_buffer_copy(group_id) if group_id is not None else None
        h = C.c_void_p()
        check(self.lib, self.lib.lumen_mi_group_create(renderer.h, rank, world, idbuf, C.byref(transport.struct) if transport is not None else None, C.byref(h)))
        self.h = h
        p = capi.TilePlan()
        check(self.lib, self.lib.lumen_mi_group_get_plan(self.h, C.byref(p)))
        self.tile, self.window, self.grid = tuple(p.tile), tuple(p.window), (p.cols, p.rows)
        self.width, self.height = renderer.GetRenderResolution()

    def close(self):
        if self.h:
            self.lib.lumen_mi_group_destroy(self.h)
            self.h = None

    def SelfTest(self):
        ms = C.c_float(0)
        check(self.lib, self.lib.lumen_mi_group_self_test(self.h, C.byref(ms)))
        return ms.value

    def TraceFrame(self): check(self.lib, self.lib.lumen_mi_group_trace_frame(self.h))
    def Gather(self): check(self.lib, self.lib.lumen_mi_group_gather(self.h))
    def Synchronize(self): check(self.lib, self.lib.lumen_mi_group_synchronize(self.h))

    def GetFrame(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        check(self.lib, self.lib.lumen_mi_group_get_frame(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.nbytes))
        return out

    def FrameDevicePointer(self):
        p = C.c_void_p()
        check(self.lib, self.lib.lumen_mi_group_frame_device(self.h, C.byref(p)))
        return p.value

    def Stats(self):
        n, ms = C.c_uint64(0), C.c_float(0)
        check(self.lib, self.lib.lumen_mi_group_get_stats(self.h, C.byref(n), C.byref(ms)))
        return {"gathers": n.value, "mean_gather_ms": ms.value}
