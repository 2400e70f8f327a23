"""One rank of the tiled multi-GPU path on real GPUs (backend nccl = RCCL; with LUMEN_WORKER_ONE_GPU=1 every rank renders on GPU 0 and the same
operations travel over gloo through host staging: the multi-PROCESS path — rendezvous, halo plan, exchange order, gather — with real rendering on a 1-GPU box).  Launched by tests/test_gpu_parity.py through
torch.distributed.run when the box has at least two GPUs:  python -m torch.distributed.run --nproc-per-node N tests/multigpu_worker.py
Every rank renders its window (tile + 60-px halo) of the Cornell box at an odd depth (temporal history is live, so the seam exchange of
the halo rings' reservoirs runs after every TraceFrame), the tiles are gathered on rank 0 with one RCCL gather per displayed frame, and
rank 0 compares every blended frame bit for bit with its own single-GPU render of the full image."""
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
    from lumenrenderer_amd import tiles
    one_gpu = os.environ.get("LUMEN_WORKER_ONE_GPU") == "1"
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    W, H, D, frames = 320, 256, 5, 4
    d = cornell()
    tile = tiles.tile_rect(rank, world, W, H); win = tiles.window_rect(tile, W, H)
    if one_gpu:
        dist.init_process_group("gloo")
        from host_staged_dist import HostStagedDist
        real_dist, dist = dist, HostStagedDist(dist)          # device tensors staged through the host: RCCL refuses two ranks on one device
    else:
        dist.init_process_group("nccl", device_id=dev)
        real_dist = dist
    r = product_from(d, W, H, D, blend=True, window=win, device=local_rank)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.SetTile(*tile)
    hx = tiles.HistoryExchange(r, rank, world, W, H, dev) if world > 1 else None     # from the first frame on: every frame's history crosses the seams
    gather = tiles.TileGather(rank, world, W, H, torch.float32, dev)
    full = product_from(d, W, H, D, blend=True, device=local_rank) if rank == 0 else None
    bad = 0
    for f in range(frames):
        r.TraceFrameAsync()
        if hx is not None:
            hx.run(dist)
        img = gather.run_renderer(r, dist)                       # tile -> send buffer -> ONE gather -> assembled frame on rank 0 (pitched copies by the library's own kernel)
        if rank == 0:
            assert full.TraceFrame() is True
            want = full.GetRadiance()
            got = img.cpu().numpy()
            mism = int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
            print(f"frame {f}: {mism} differing words", flush=True)
            bad += mism
    verdict = torch.tensor([bad], dtype=torch.int64, device="cpu" if one_gpu else dev)
    real_dist.broadcast(verdict, src=0)
    r.close()
    if full is not None:
        full.close()
    real_dist.destroy_process_group()
    if int(verdict[0]) != 0:
        raise SystemExit(f"rank {rank}: stitched frames differ from the single-GPU render")
    if rank == 0:
        print("MULTIGPU OK", flush=True)


if __name__ == "__main__":
    main()
