"""One rank of the NATIVE tile group (include/lumen_mi.h "tile groups", csrc/group.cpp) as a process: plan, seam exchange, double-buffered gather all run in C++ behind the C ABI.
torch.distributed is used for two things only: handing rank 0's communicator id to the other ranks (a broadcast of 256 bytes over gloo) and the final verdict.  Transport:
RCCL over xGMI when every rank has its own GPU; with LUMEN_WORKER_ONE_GPU=1 every rank renders on GPU 0 and the group gets a HOST transport (lumenrenderer_amd.group.DistHostTransport:
gloo point-to-point on the pinned staging buffers of the group) — RCCL refuses several ranks per device.  Launched by tests/test_zz_multiprocess.py:
python -m torch.distributed.run --nproc-per-node N tests/group_worker.py.  Cornell box at an odd depth (temporal history live: the seam exchange runs after every frame), blended
frames; rank 0 renders the full image beside it and compares every gathered frame bit for bit.  The gather of frame f is read only after frame f + 1 has been enqueued (the overlap the
double buffering exists for)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from helpers import cornell, product_from
    from lumenrenderer_amd import group
    one_gpu = os.environ.get("LUMEN_WORKER_ONE_GPU") == "1"
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    W, H, D, frames = 320, 256, int(os.environ.get("LUMEN_WORKER_DEPTH", "5")), 5
    dist.init_process_group("gloo")
    d = cornell()
    r = product_from(d, W, H, D, blend=True, device=local_rank)
    if one_gpu:
        transport, gid = group.DistHostTransport(dist), None
    else:
        transport = None
        box = [group.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        gid = box[0]
    g = group.TileGroup(r, rank, world, group_id=gid, transport=transport)
    ms = g.SelfTest()
    print(f"rank {rank}: tile {g.tile} window {g.window} grid {g.grid} self-test {ms:.1f} ms", flush=True)
    full = product_from(d, W, H, D, blend=True, device=local_rank) if rank == 0 else None
    bad = 0
    want = []
    for f in range(frames):                        # pipelined: nothing waits between the frames, gather f runs beside frame f + 1
        g.TraceFrame()
        g.Gather()
        if rank == 0:
            assert full.TraceFrame() is True
            want.append(full.GetRadiance())
    # read back: the LAST gathered frame must be the last blended frame; earlier frames are checked by a second pass that reads after every gather
    g.Synchronize()
    if rank == 0:
        got = g.GetFrame()
        mism = int(np.sum(got.view(np.uint32) != want[-1].view(np.uint32)))
        print(f"pipelined pass, last frame: {mism} differing words", flush=True)
        bad += mism
    for f in range(3):                             # frame by frame
        g.TraceFrame(); g.Gather()
        if rank == 0:
            assert full.TraceFrame() is True
            got, ref = g.GetFrame(), full.GetRadiance()
            mism = int(np.sum(got.view(np.uint32) != ref.view(np.uint32)))
            print(f"frame {frames + f}: {mism} differing words", flush=True)
            bad += mism
        else:
            g.Synchronize()
    st = g.Stats()
    print(f"rank {rank}: {st}", flush=True)
    if transport is not None and transport.errors:
        print("transport errors:", transport.errors[:3], flush=True)
        bad += 1
    verdict = torch.tensor([bad], dtype=torch.int64)
    dist.broadcast(verdict, src=0)
    g.close(); r.close()
    if full is not None:
        full.close()
    dist.destroy_process_group()
    if int(verdict[0]) != 0:
        raise SystemExit(f"rank {rank}: stitched frames differ from the single-GPU render")
    if rank == 0:
        print("GROUP OK", flush=True)


if __name__ == "__main__":
    main()
