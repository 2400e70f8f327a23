"""Several ranks on ONE GPU: the transport shim behind the multi-process rehearsals of the tiled path on the 1-GPU boxes of the development pool
(tests/multigpu_worker.py with LUMEN_WORKER_ONE_GPU=1, bench.py with LUMEN_BENCH_ONE_GPU=1).  Test infrastructure: the product (lumenrenderer_amd/) never imports it."""


class HostStagedDist:
    """A ``torch.distributed`` look-alike for ranks that SHARE one GPU: RCCL refuses two ranks on a device and gloo does not move device tensors of one GPU
    between processes, so the tensor operations of this module and of bench.py (all_reduce, all_gather, gather, batch_isend_irecv of P2POp(isend / irecv)) are staged
    device -> host -> gloo -> device; everything else (barrier, all_gather_object, get_world_size, ...) is the real module's.  ``.cpu()`` / ``copy_`` run on torch's
    current stream — the renderer's — so the ordering is the production one.  Not a transport anybody should measure: tests/multigpu_worker.py and
    ``LUMEN_BENCH_ONE_GPU=1 bench.py --gpus N`` use it to EXECUTE the N-rank code paths where only one GPU exists."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Request:
        def __init__(self, req, host, dev):
            self.req, self.host, self.dev = req, host, dev

        def wait(self):
            self.req.wait()
            if self.dev is not None:
                self.dev.copy_(self.host)

    isend, irecv = "isend", "irecv"

    def __init__(self, dist):
        self._d = dist

    def __getattr__(self, name):                          # barrier, get_world_size, is_initialized, all_gather_object, ReduceOp, destroy_process_group, ...
        return getattr(self._d, name)

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h) if op is None else self._d.all_reduce(h, op=op)
        t.copy_(h)

    def all_gather(self, outs, t):
        import torch
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        self._d.all_gather(hs, t.cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)

    def gather(self, send, parts, dst=0):
        import torch
        hp = [torch.empty(p.shape, dtype=p.dtype) for p in parts] if parts else None
        self._d.gather(send.contiguous().cpu(), hp, dst=dst)
        if parts:
            for p, h in zip(parts, hp):
                p.copy_(h)

    def batch_isend_irecv(self, ops):
        import torch
        real, pairs = [], []
        for o in ops:
            send = o.op == "isend"
            h = o.tensor.cpu() if send else torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
            real.append(self._d.P2POp(self._d.isend if send else self._d.irecv, h, o.peer))
            pairs.append((h, None if send else o.tensor))
        return [HostStagedDist._Request(r, h, t) for r, (h, t) in zip(self._d.batch_isend_irecv(real), pairs)]
