#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: Mrays/s and ms/frame at 1440p, 4 spp, depth 6, Sponza (stand-in).

A "step" is one displayed frame = 4 blended TraceFrame() calls (the reference has no spp parameter: SURVEY.md F3) over
synthetic geometry already resident in HBM.  N GPUs shard the frame by tile (lumenrenderer_amd/tiles.py) and gather the
radiance on rank 0 with one RCCL collective; the total work is fixed, so scaling is "strong".
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The frame graph keeps four HIP streams busy; the RCCL communicator of a multi-GPU run adds its own.  HIP multiplexes streams
# onto 4 hardware queues by default, and two busy streams sharing a queue serialise (measured: -11 %): ask for 8 before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: (scene factory kwargs, width, height, depth, spp)
    "c2": ("sponza", dict(), 2560, 1440, 6, 4),
    "c3": ("sponza", dict(extra_lights=512), 2560, 1440, 6, 4),
    "c4": ("sponza", dict(), 3840, 2160, 8, 8),
    "c5": ("foliage", dict(), 1920, 1080, 6, 1),
    "c1": ("cornell", dict(), 256, 256, 2, 1),
}


def make_scene(kind, kw):
    from lumenrenderer_amd import scenes
    if kind == "sponza":
        return scenes.sponza_standin(**kw)
    if kind == "foliage":
        return scenes.foliage_stress(**kw)
    return scenes.cornell_box(fixture=os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))


def algorithmic_bytes_closest(rays, nodes, tris):
    """closest-hit launch = 56 B/ray (read 40 + write 16) + 64 B per BVH2 node visited + 48 B per Woop packet tested."""
    return 56.0 * rays + 64.0 * nodes + 48.0 * tris


def algorithmic_bytes_traceframe(c, depth, npix, nodes_all, tris_all, blend=True):
    """SURVEY.md §8 d4, whole TraceFrame: the fixed per-event accounting of the reference's data flow (AoS-equivalent minimum
    traffic), independent of how many bytes this implementation really moves.  c = lumen_mi_get_counters()."""
    waves = [c[4 + d] for d in range(depth)]
    closest, shadow, restir = c[0], c[1], c[2]
    b = 40.0 * npix                                         # primary generation
    b += 56.0 * closest + 420.0 * closest                   # closest-hit launches + surface extraction (one per traced ray)
    live = sum(waves[:max(0, depth - 1)])                   # path vertices that go through ShadeIndirect (depth < maxDepth-1)
    emitted = sum(waves[1:depth])
    b += 176.0 * live + 40.0 * emitted                      # ShadeIndirect: read 176, write 40 when the path survives
    b += 300.0 * sum(waves[1:depth])                        # ShadeDirect at depth >= 1
    b += 64.0 * shadow                                      # NEE shadow launches
    b += 5196.0 * npix + 180.0 * npix                       # ReSTIR passes + motion vectors
    b += (48.0 if blend else 40.0) * npix                   # merge
    b += 64.0 * nodes_all + 48.0 * tris_all                 # traversal of every ray type (measured node / triangle visits)
    return b


def load_traffic():
    """HBM bytes per launch from PMC counters (separate rocprofv3 --pmc passes, tools/traffic.sh); None when not measured."""
    path = os.path.join(ROOT, "profiles", "r01_c2_hbm_traffic_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None


def cpu_baseline(kind, kw, depth, spp, full):
    """The CPU oracle ("port") timed on this box's host cores on a bounded sample of the same workload: the same scene,
    `spp` blended frames, at the largest of a few resolutions expected to need <= ~25 s (probed at 480x270 first).
    Test infrastructure used as a reported baseline only — never as the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from
    from oracle_lib import usable_cpus
    cores = usable_cpus()                                 # affinity mask capped by the cgroup CPU quota: the threads that really run
    scene = make_scene(kind, kw)

    def run(w, h):
        o = oracle_from(scene, w, h, depth, blend=True, threads=cores)
        o.world_triangles()                               # scene flattening + BVH build outside the timed region (as on the GPU)
        t0 = time.perf_counter()
        rays = 0
        for _ in range(spp):
            o.trace_frame()
            s = o.stats(4)
            rays += s[0] + s[1] + s[2]
        dt = time.perf_counter() - t0
        o.close()
        return rays, dt

    w, h = 480, 270
    rays, dt = run(w, h)
    for cw, ch in ((full[0], full[1]), (1920, 1080), (1280, 720), (960, 540)):
        if cw * ch <= full[0] * full[1] and dt * (cw * ch) / (480 * 270) <= 25.0 and (cw, ch) != (w, h):
            w, h = cw, ch
            rays, dt = run(w, h)
            break
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{w}x{h} x {spp} blended frames, depth {depth}, same scene: {dt:.1f} s, {rays} rays, {dt * 1e3:.0f} ms/frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-seams", choices=["auto", "on", "off"], default="auto",
                    help="multi-GPU: exchange the halo rings' reservoir history (and the executed-wave count) after every TraceFrame; "
                         "auto = only for path depths that have temporal history (odd number of waves per frame)")
    ap.add_argument("--emulate-rank", default="", help="R/N: on ONE GPU render only rank R's window of an N-GPU tile grid (no gather); "
                    "design aid for the per-rank time of the tiled path, never the reported benchmark line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lumenrenderer_amd import LumenRendererMI, tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if os.environ.get("LUMEN_BENCH_FORCE_PG", "") == "before":      # A/B aid: communicator (and its stream) BEFORE the renderer's streams
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        warm = torch.zeros(1, device=dev); dist.all_reduce(warm); torch.cuda.synchronize()
    kind, kw, W, H, depth, spp = WORKLOADS[args.workload]
    desc = make_scene(kind, kw)
    r = LumenRendererMI()
    r.Init(depth=depth, render_resolution=(W, H), blend_output=True, device=local_rank)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.LoadSceneDescription(desc)
    emu = tuple(int(x) for x in args.emulate_rank.split("/")) if args.emulate_rank else None
    if emu and world != 1:
        raise SystemExit("--emulate-rank is a single-GPU design aid")
    tile = tiles.tile_rect(*emu, W, H) if emu else tiles.tile_rect(rank, world, W, H)
    win = tiles.window_rect(tile, W, H) if (world > 1 or emu) else (0, 0, W, H)
    r.SetWindow(*win)
    if world > 1 or emu:
        r.SetTile(*tile)                                  # halo pixels only get the work the tile's ReSTIR reuse needs
    wh, ww = win[3] - win[1], win[2] - win[0]
    window_buf = torch.empty((wh, ww, 4), dtype=torch.float32, device=dev)
    # The renderer's four streams are created and used once BEFORE the RCCL communicator brings its own stream: HIP maps
    # streams onto 4 hardware queues, and two busy streams that end up sharing one serialise (measured -11 % at N = 1 with an
    # idle fifth stream created first).  RCCL's stream only works between frames, when the renderer's streams are idle.
    r.SetBlendMode(True)
    r.TraceFrame()
    force_pg = os.environ.get("LUMEN_BENCH_FORCE_PG", "")          # A/B aid: "before" / "after" create a 1-rank communicator at N = 1
    if (world > 1 or force_pg) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        warm = torch.zeros(1, device=dev); dist.all_reduce(warm)                                   # communicator + stream exist now
        torch.cuda.synchronize()

    # temporal history across tile seams: only a path depth with an odd number of waves per frame has any (tiles.history_needed);
    # then every TraceFrame is followed by one grouped RCCL send / recv of the halo rings' reservoirs
    exact = args.exact_seams == "on" or (args.exact_seams == "auto" and tiles.history_needed(depth))
    hx = tiles.HistoryExchange(r, rank, world, W, H, dev) if (world > 1 and exact) else None

    def frame():
        r.SetBlendMode(True)                              # a fresh 4-spp accumulation per displayed frame
        for _ in range(spp):
            r.TraceFrameAsync()
            if hx is not None:
                hx.run(dist)
        r.CopyRadianceToDevice(window_buf.data_ptr())
        local = window_buf[tile[1] - win[1]: tile[3] - win[1], tile[0] - win[0]: tile[2] - win[0]]
        return local if emu else tiles.gather_tiles(local, rank, world, W, H, dist)

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented pass (outside the timed region): BVH nodes / triangles per closest-hit ray for the roofline
    r.SetInstrumented(True)
    r.SetBlendMode(True)
    r.TraceFrame()
    ci = r.GetCounters()
    nodes_per_ray = ci[20] / max(1, ci[0] + ci[1] + ci[2])
    tris_per_ray = ci[21] / max(1, ci[0] + ci[1] + ci[2])
    r.SetInstrumented(False)

    for _ in range(args.warmup):
        frame()
    r.EnableKernelTiming(True)
    barrier()
    t0 = time.perf_counter()
    rays_total = 0
    closest_ms, closest_launches, closest_rays, shadow_ms, shade_ms, restir_ms = 0.0, 0, 0, 0.0, 0.0, 0.0
    img = None
    for _ in range(args.steps):
        img = frame()
    barrier()
    dt = time.perf_counter() - t0
    r.EnableKernelTiming(False)
    # counters of the LAST TraceFrame (every TraceFrame of a step traces the same number of rays to within RNG noise);
    # kernel times are HIP-event sums over the whole timed region, on the stream the kernels were launched on
    c = r.GetCounters(50)
    rays_last = c[0] + c[1] + c[2]
    closest_ms, closest_launches = r.GetKernelTime(0)
    shadow_ms, _ = r.GetKernelTime(1)
    shade_ms, _ = r.GetKernelTime(2)
    restir_ms, _ = r.GetKernelTime(3)
    total_ms, n_traceframes = r.GetKernelTime(4)
    n_traceframes = max(1, n_traceframes)

    # a rank's counters include the rays of its halo pixels; those are redundant work (the neighbour owns the pixels), so
    # only the tile's share is counted: rays scale with pixels to within RNG noise
    tile_share = ((tile[2] - tile[0]) * (tile[3] - tile[1])) / float(ww * wh)
    # primary rays and the first ReSTIR visibility pass cover the whole window (scaled to the tile); indirect waves, NEE and
    # the second visibility pass only run for tile pixels already (lumen_mi_set_tile)
    rays_tile = (c[4] + c[48]) * tile_share + (c[0] - c[4]) + c[1] + c[49] if (world > 1 or emu) else float(rays_last)
    stats = torch.tensor([dt, float(rays_tile)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); rays_last_all = float(tsum[1])
    else:
        rays_last_all = float(rays_tile)
    if rank == 0:
        # rays/frame: the spp TraceFrame()s of one step trace (to within RNG noise) the same number of rays each
        rays_per_frame = rays_last_all * spp
        ms_per_step = dt * 1e3 / args.steps
        value = rays_per_frame / (ms_per_step * 1e-3) / 1e6
        # roofline of the dominant kernel (closest-hit traversal) on rank 0: algorithmic bytes / measured device time
        nodes_c = ci[20] * (c[0] / max(1, ci[0] + ci[1] + ci[2]))       # share of instrumented counts attributed to closest-hit rays
        tris_c = ci[21] * (c[0] / max(1, ci[0] + ci[1] + ci[2]))
        alg = algorithmic_bytes_closest(c[0], nodes_c, tris_c)              # per TraceFrame (all `depth` closest-hit launches)
        launches_per_tf = closest_launches / n_traceframes
        per_launch_ms = closest_ms / max(1, closest_launches)
        achieved = (alg / max(1.0, launches_per_tf)) / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        traffic = load_traffic() if args.workload == "c2" and world == 1 else None
        traffic_closest = traffic.get("lm_k_trace_closest", {}).get("hbm_bytes_per_launch_corrected") if traffic else None
        alg_tf = algorithmic_bytes_traceframe(c, depth, (win[2] - win[0]) * (win[3] - win[1]), ci[20], ci[21])
        out = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {'cornell box (the reference asset, tests/golden fixture)' if kind == 'cornell' else kind + ' stand-in'}, {W}x{H}, {spp} spp (blended TraceFrames), depth {depth}, ReSTIR DI on",
                       "triangles": desc.triangle_count(), "rays_per_frame": int(rays_per_frame), "ms_per_frame": round(ms_per_step, 4),
                       "tiles": f"{tiles.grid_for(world, W, H)[0]}x{tiles.grid_for(world, W, H)[1]} + {tiles.HALO}px halo" if world > 1 else "1x1",
                       "nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2),
                       "rays_per_wave": [int(c[4 + d]) for d in range(depth)], "nee_shadow_rays": int(c[1]), "restir_shadow_rays": int(c[2]),
                       "algorithmic_bytes_per_traceframe": int(alg_tf),
                       "hbm_fraction_by_algorithmic_bytes": round(alg_tf * spp / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) if world == 1 else None},
            "roofline": {"bound": "hbm", "kernel": "lm_k_trace_closest", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_closest,
                         "launch_ms": round(per_launch_ms, 4), "launches_per_traceframe": launches_per_tf,
                         "algorithmic_bytes_per_launch": int(alg / max(1.0, launches_per_tf))},
            "device_ms_per_traceframe": {"closest": round(closest_ms / n_traceframes, 3), "shadow": round(shadow_ms / n_traceframes, 3),
                                         "shade": round(shade_ms / n_traceframes, 3), "restir": round(restir_ms / n_traceframes, 3),
                                         "total": round(total_ms / n_traceframes, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, kw, depth, spp, (W, H))
        if emu:
            out["emulated_rank"] = {"rank": emu[0], "of": emu[1], "window": list(win), "tile": list(tile),
                                    "note": "per-rank time of the tiled path on one GPU; value counts the tile's rays only"}
        try:                                   # RCCL writes a version banner through C stdio: flush it first so that the JSON is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
