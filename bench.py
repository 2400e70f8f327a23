#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: Mrays/s and ms/frame at 1440p, 4 spp, depth 6, Sponza (stand-in).

A "step" is one displayed frame = 4 blended TraceFrame() calls (the reference has no spp parameter: SURVEY.md F3) over
synthetic geometry already resident in HBM.  N GPUs shard the frame by tile and gather the radiance on rank 0 over RCCL; the total work is fixed, so
scaling is "strong".  The N-GPU frame path is the tile group of the C ABI (csrc/group.cpp behind lumen_mi_group_*: plan, seam exchange, double-buffered gather
on its own stream, in C++; --transport native, the default) — torch.distributed over gloo then only carries rank 0's communicator id, the barriers of the timed
region and the statistics; --transport torch is rounds 1 - 5's lumenrenderer_amd/tiles.py over torch.distributed (backend nccl).
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.

Arithmetic mode: `value` is measured with the fast ReSTIR mode (tuning key fast_resample: hardware rcp / rsq / sqrt and the contracted target
function in the candidate pick and the reuse passes; radiance within 1e-8 relative L2 of the oracle on every BASELINE configuration, 1e-3
asserted by tests/test_gpu_parity.py, every ray counter identical).  The same run then times the exact mode (bit-identical to the oracle) the
same way and reports it as config.other_mode; --mode exact swaps the two, --no-exact skips the second pass.

History passes: `value` is measured with BOTH spatial reuse passes and the reservoir combine launched in every TraceFrame, as the reference launches
them (tuning key lazy_reuse 0).  The renderer's default at this even path depth is lazy reuse — those passes only build the history of the next
frame, which the reference's own swap quirk never reads at an even depth, so they run only when the result can be read; every image, counter and ray
is identical (tests/test_gpu_parity.py::test_history_passes_run_only_when_their_result_can_be_read).  That rate is reported beside the headline as
`value_lazy_reuse` / `ms_per_step_lazy_reuse` (and `value_exact_lazy_reuse`); --reuse lazy makes it the headline instead.
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
# fp32 vector issue peak in lane-operations (one per lane per VALU instruction, an FMA = 1): 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz =
# 78.6 T lane-ops/s = 1 229 G wave instructions/s (a wave64 v_fma_f32 takes 2 cycles on a CDNA4 SIMD, MI355X_MICROARCH.md; 157.3 TFLOPS =
# this x 2 for FMA).  Measured on the box with tools/valu_peak.hip -> profiles/r03_valu_peak.txt.  (Round 2 divided by 39.3: wrong by 2x.)
VALU_PEAK_TLANEOPS = 78.6432
WAVEINST_PEAK = 1055.8e9         # VALU wave-instructions/s the chip sustains with dependent chains at 8 waves per SIMD (measured, profiles/r03_valu_peak.txt: 2.33 cycles per instruction per SIMD)
TRAV_T0_US, TRAV_STEPS_PER_US = 60.0, 75000.0  # closest-hit launch of incoherent rays: T = T0 + rays x steps per ray / S; calibration 4.6 G rays/s at 16.3 steps per ray (profiles/r04_step_latency.txt)
PMC_FILE = os.path.join("profiles", "r06_c2_pmc.json")      # tools/pmc_json.sh on the GPU box; replayed here, never measured by this run


def kernel_source_id():
    """sha256 over the device sources: the PMC replay is only valid for the build it was measured on (tools/pmc_json.sh stores this id)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lumenrenderer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

WORKLOADS = {
    # name: (scene factory kwargs, width, height, depth, spp)
    "c2": ("sponza", dict(), 2560, 1440, 6, 4),
    "c2t": ("sponza", dict(textured=True), 2560, 1440, 6, 4),        # C2 with seeded 1024^2 base-colour / normal / metal-roughness maps on 22 materials
    "c3": ("sponza", dict(extra_lights=512), 2560, 1440, 6, 4),
    "c4": ("sponza", dict(), 3840, 2160, 8, 8),
    "c5": ("foliage", dict(), 1920, 1080, 6, 1),
    "c1": ("cornell", dict(), 256, 256, 2, 1),
    # the reference's OWN default setting (Sandbox/src/Application.cpp:89-93: 1280x720, depth 5, ReSTIR on) with the camera moving every TraceFrame, which
    # switches blending off (OutputLayer.cpp:492-495): 1 "spp", an odd depth — the reservoir swap chain turns every frame, the temporal pass reads a live
    # history through non-zero motion vectors, and the history passes run with their frame (no lazy reuse).  A step = one TraceFrame at the next camera pose.
    "sandbox": ("sponza", dict(), 1280, 720, 5, 1),
    # the same setting on the reference's own DEFAULT MODEL (Sandbox/src/AppConfigDefaults.h:11: LowpolyRoom/scene.glb, 20 501 triangles, lit by its three emissive
    # materials only = 414 triangle lights; tests/golden/ref_lowpoly_room.npz) from the camera Application.cpp:145-146 sets, walking as OutputLayer.cpp's input handling
    # moves the reference's own Camera class (W held + mouse drag; 64 poses in the fixture)
    "lowpoly": ("lowpoly", dict(), 1280, 720, 5, 1),
}
MOVING = {"sandbox", "lowpoly"}             # workloads whose camera moves every TraceFrame (lumenrenderer_amd.scenes.sandbox_camera_pose), blending off


def make_scene(kind, kw):
    from lumenrenderer_amd import scenes
    if kind == "sponza":
        return scenes.sponza_standin(**kw)
    if kind == "foliage":
        return scenes.foliage_stress(**kw)
    if kind == "lowpoly":
        return scenes.lowpoly_room(os.path.join(ROOT, "tests", "golden", "ref_lowpoly_room.npz"))
    return scenes.cornell_box(fixture=os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))


def algorithmic_bytes_closest(rays, nodes, tris):
    """Closest-hit launch: 56 B/ray (read 40 + write 16) + 64 B per node record fetched + 48 B per Woop packet tested.
    `nodes` = 64-byte 4-wide node records the traversal really fetches (16.5 per ray on C2).  SURVEY d4 prices a BINARY node at 64 B;
    the same visits expressed in binary nodes are twice as many (two child boxes each), which is the figure `*_d4` fields carry."""
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


def load_pmc():
    """Per-kernel PMC figures (separate rocprofv3 --pmc passes of tools/pmc_json.sh, corrected as MI355X_MICROARCH.md prescribes), taken
    on the GPU box by the builder and committed under profiles/: a REPLAY of that measurement, not a measurement of this run — and only
    of the SAME device sources (the file carries kernel_source_id()); a stale file is refused and the PMC-derived fields are null."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            j = json.load(f)
        if j.get("kernel_source_id") != kernel_source_id():
            return {}
        return j.get("kernels", {})
    except (OSError, ValueError):
        return {}


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run` over this script as a CHILD process (never an
    exec: nothing here may replace a process, and nothing in this branch touches the GPU), pass its output through with rank 0's JSON
    line last, and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    # LUMEN_BENCH_LOG_DIR: every rank's stdout / stderr also lands in <dir>/<run>/attempt_0/<rank>/{stdout,stderr}.log, so that a rank that dies can be read afterwards
    logs = ["--log-dir", os.environ["LUMEN_BENCH_LOG_DIR"], "--tee", "3"] if os.environ.get("LUMEN_BENCH_LOG_DIR") else []
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + logs + [os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    if args.dry_launch:
        # the launch command and what every rank would render and send: its tile, its window (tile + 60-px halo, clipped), the redundant halo pixels, the bytes it
        # contributes to the one gather per displayed frame and, for path depths with temporal history, the seam exchange after every TraceFrame — without touching a GPU
        from lumenrenderer_amd import tiles
        _, _, W, H, depth, spp = WORKLOADS[args.workload]
        cols, rows = tiles.grid_for(args.gpus, W, H)
        mh, mw = tiles.max_tile_shape(args.gpus, W, H)
        ranks = []
        for rk in range(args.gpus):
            t = tiles.tile_rect(rk, args.gpus, W, H); w = tiles.window_rect(t, W, H)
            tp, wp = (t[2] - t[0]) * (t[3] - t[1]), (w[2] - w[0]) * (w[3] - w[1])
            plan = tiles.halo_plan(rk, args.gpus, W, H) if tiles.history_needed(depth) else []
            area = lambda q: 0 if q is None else (q[2] - q[0]) * (q[3] - q[1])
            ranks.append({"rank": rk, "tile": list(t), "window": list(w), "tile_pixels": tp, "window_pixels": wp, "halo_pixels": wp - tp, "halo_over_tile": round((wp - tp) / tp, 4),
                          "gather_send_bytes": mh * mw * 16, "seam_send_bytes_per_traceframe": sum(area(sd) for _, sd, _ in plan) * tiles.HISTORY_FLOATS * 4,
                          "seam_recv_bytes_per_traceframe": sum(area(rv) for _, _, rv in plan) * tiles.HISTORY_FLOATS * 4, "seam_peers": [p_ for p_, _, _ in plan]})
        print(json.dumps({"launch": cmd, "workload": args.workload, "image": [W, H], "depth": depth, "spp": spp, "grid": f"{cols}x{rows}", "halo_px": tiles.HALO,
                          "gather": {"collective": "gather to rank 0 (RCCL over xGMI), one per displayed frame", "bytes_per_rank": mh * mw * 16, "bytes_total": args.gpus * mh * mw * 16},
                          "seam_exchange": "after every TraceFrame (odd path depth: temporal history is live)" if tiles.history_needed(depth) else "none (even path depth: no temporal history)",
                          "worst_window_pixels": max(r_["window_pixels"] for r_ in ranks), "single_gpu_pixels": W * H, "ranks": ranks}))
        return 0
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import re
    import threading
    for attempt in range(3):
        # the rendezvous port was free a moment ago; on a host that other jobs share it can be taken before the launcher listens on it (EADDRINUSE): new port, again
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        etext = []
        def pump():                                             # the child's stderr passes through as it comes (a watchdog message must not wait for the end) and is kept
            for eline in p.stderr:
                etext.append(eline); sys.stderr.write(eline); sys.stderr.flush()
        t = threading.Thread(target=pump, daemon=True); t.start()
        line_json = None
        for line in p.stdout:
            s = line.strip()
            m = re.match(r"^\[[A-Za-z_]+\d+\]:(.*)$", s)      # --tee prefixes every line with its rank ("[default0]:")
            if m and m.group(1).startswith("{") and '"metric"' in m.group(1):
                s = m.group(1)
            if s.startswith("{") and '"metric"' in s:
                line_json = s                                   # held back: printed last
            else:
                sys.stdout.write(line); sys.stdout.flush()
        rc = p.wait(); t.join(timeout=10)
        err = "".join(etext)
        if rc != 0 and line_json is None and attempt < 2 and ("EADDRINUSE" in err or "address already in use" in err.lower()):
            with socket.socket() as s2:
                s2.bind(("127.0.0.1", 0)); port = s2.getsockname()[1]
            cmd[cmd.index("--master-port") + 1] = str(port)
            sys.stderr.write(f"bench.py: rendezvous port taken by another process, retrying on {port}\n")
            continue
        break
    if line_json is not None:
        print(line_json, flush=True)
    return rc


def camera_pose(desc, k):
    """Pose of the k-th TraceFrame of a moving workload: the 32-frame walk of scenes.sandbox_camera_pose there and back again (the camera stays in the atrium
    however many steps are timed; every frame still moves by one step)."""
    from lumenrenderer_amd import scenes
    if hasattr(desc, "camera_poses"):                     # LowpolyRoom: the fixture's 64 poses of the reference's Camera class, there and back
        k %= 126
        return scenes.lowpoly_camera_pose(desc, k if k < 64 else 126 - k)
    k %= 64
    return scenes.sandbox_camera_pose(desc, k if k < 32 else 64 - k)


def cpu_model():
    """Model string of the host CPU (BASELINE.md §3 asks for model + nproc + threads beside every CPU number)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(kind, kw, depth, spp, full, moving=False):
    """The CPU oracle ("port") timed on this box's host cores on a bounded sample of the same workload: the same scene,
    `spp` blended frames, at the largest of a few resolutions expected to need <= ~25 s (probed at 480x270 first).
    Test infrastructure used as a reported baseline only — never as the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_from
    from oracle_lib import usable_cpus
    cores = usable_cpus()                                 # affinity mask capped by the cgroup CPU quota: the threads that really run
    scene = make_scene(kind, kw)

    frames = 4 if moving else spp                       # a moving workload: four consecutive poses (one frame alone has no history to read)

    def run(w, h):
        o = oracle_from(scene, w, h, depth, blend=not moving, threads=cores)
        o.world_triangles()                               # scene flattening + BVH build outside the timed region (as on the GPU)
        t0 = time.perf_counter()
        rays = 0
        for k in range(frames):
            if moving:
                o.set_camera(*camera_pose(scene, k))
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
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "threads": cores, "cpu_model": cpu_model(), "nproc": os.cpu_count(), "kind": "port",
            "sample": (f"{w}x{h} x {frames} TraceFrames at consecutive camera poses, blending off" if moving else f"{w}x{h} x {spp} blended frames") +
                      f", depth {depth}, same scene: {dt:.1f} s, {rays} rays, {dt * 1e3 / (frames if moving else 1):.0f} ms/frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast",
                    help="arithmetic of the ReSTIR target function for the headline number: 'fast' = hardware rcp / rsq / sqrt + contracted "
                         "evaluation (tuning key fast_resample; held to 1e-3 rel-L2 against the oracle by tests/test_gpu_parity.py, measured 1e-8), "
                         "'exact' = correctly rounded, bit-identical to the oracle")
    ap.add_argument("--reuse", choices=("eager", "lazy"), default="eager",
                    help="ReSTIR history passes of the headline: eager = launched with every TraceFrame like the reference's (lazy_reuse 0); lazy = the renderer's "
                         "default, launched when their result can be read (identical images; reported beside the headline either way)")
    ap.add_argument("--no-other-reuse", action="store_true", help="skip the timed passes with the other setting of --reuse")
    ap.add_argument("--no-exact", action="store_true", help="skip the second timed pass in exact mode (reported beside the headline)")
    ap.add_argument("--exact-seams", choices=["auto", "on", "off"], default="auto",
                    help="multi-GPU: exchange the halo rings' reservoir history (and the executed-wave count) after every TraceFrame; "
                         "auto = only for path depths that have temporal history (odd number of waves per frame)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: no HIP events around the kernels in the timed region (the roofline "
                    "entry then has no live launch time)")
    ap.add_argument("--transport", choices=["native", "torch"], default=os.environ.get("LUMEN_BENCH_TRANSPORT", "native"),
                    help="multi-GPU: native = the tile group of the C ABI (csrc/group.cpp: plan, seam exchange and the double-buffered gather in C++, RCCL resolved by the library; "
                         "torch.distributed over gloo only carries the communicator id, the barriers and the statistics); torch = rounds 1 - 5: tiles.py over torch.distributed (backend nccl)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no launcher environment: print the child command as JSON and exit")
    ap.add_argument("--emulate-rank", default="", help="R/N: on ONE GPU render only rank R's window of an N-GPU tile grid (no gather); "
                    "design aid for the per-rank time of the tiled path, never the reported benchmark line")
    args = ap.parse_args()

    # --gpus N > 1 started bare (no RANK / WORLD_SIZE from a launcher): become the launcher.  Before torch or any GPU call.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import faulthandler
    faulthandler.enable()                                 # a rank that dies on a signal (SIGABRT / SIGSEGV from a library) leaves its Python stack on stderr
    import torch
    import torch.distributed as dist
    from lumenrenderer_amd import LumenRendererMI, tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # LUMEN_BENCH_ONE_GPU=1: a REHEARSAL of the N-rank path where only one GPU exists — every rank renders on GPU 0, the collectives run over gloo staged through the
    # host (tests/host_staged_dist.py).  It executes every line the N-GPU run executes (windows, seam exchange, gather, per-rank statistics); its rate means nothing and
    # the JSON line says so ("rehearsal").
    one_gpu = os.environ.get("LUMEN_BENCH_ONE_GPU", "") == "1" and world > 1
    if one_gpu:
        local_rank = 0
    present = torch.cuda.device_count()                   # counting devices does not initialise the GPU
    if present < world and not one_gpu:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible GPUs, this node shows {present} (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
                         f"`python bench.py --gpus {world} --dry-launch` prints the launch command and every rank's tile plan without touching a GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if os.environ.get("LUMEN_BENCH_FORCE_PG", "") == "before":      # A/B aid: communicator (and its stream) BEFORE the renderer's streams
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        warm = torch.zeros(1, device=dev); dist.all_reduce(warm); torch.cuda.synchronize()
    if world > 1 and "LUMEN_MI_BUILD_THREADS" not in os.environ:          # N ranks build the same tree at the same time on one host: share its cores
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        try:
            from oracle_lib import usable_cpus
            os.environ["LUMEN_MI_BUILD_THREADS"] = str(max(1, usable_cpus() // world))
        except Exception:
            os.environ["LUMEN_MI_BUILD_THREADS"] = str(max(1, (os.cpu_count() or world) // world))
    native = args.transport == "native" and world > 1
    kind, kw, W, H, depth, spp = WORKLOADS[args.workload]
    moving = args.workload in MOVING
    blend_on = not moving
    desc = make_scene(kind, kw)
    r = LumenRendererMI()
    r.Init(depth=depth, render_resolution=(W, H), blend_output=blend_on, device=local_rank)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.LoadSceneDescription(desc)
    emu = tuple(int(x) for x in args.emulate_rank.split("/")) if args.emulate_rank else None
    if emu and world != 1:
        raise SystemExit("--emulate-rank is a single-GPU design aid")
    tile = tiles.tile_rect(*emu, W, H) if emu else tiles.tile_rect(rank, world, W, H)
    win = tiles.window_rect(tile, W, H) if (world > 1 or emu) else (0, 0, W, H)
    if not native:                                        # (the native group sets window and tile itself when it is created, below)
        r.SetWindow(*win)
        if world > 1 or emu:
            r.SetTile(*tile)                              # halo pixels only get the work the tile's ReSTIR reuse needs
    wh, ww = win[3] - win[1], win[2] - win[0]
    emu_tile_buf = torch.empty((tile[3] - tile[1], tile[2] - tile[0], 4), dtype=torch.float32, device=dev) if emu else None
    # The renderer's four streams are created and used once BEFORE the RCCL communicator brings its own stream: HIP maps
    # streams onto 4 hardware queues, and two busy streams that end up sharing one serialise (measured -11 % at N = 1 with an
    # idle fifth stream created first).  RCCL's stream only works between frames, when the renderer's streams are idle.
    r.SetBlendMode(blend_on)
    if native:
        r.SetWindow(*win); r.SetTile(*tile)              # the first frame (streams, buffers) at the size the group will render
    r.TraceFrame()
    force_pg = os.environ.get("LUMEN_BENCH_FORCE_PG", "")          # A/B aid: "before" / "after" create a 1-rank communicator at N = 1
    grp = None
    if native:
        # ---- the native tile group: torch.distributed (gloo, CPU) hands rank 0's communicator id round, synchronises the timed region and collects the statistics;
        # every byte of the frame path moves through csrc/group.cpp (RCCL, or — rehearsal on one GPU — a host transport over the same gloo group)
        import threading
        from lumenrenderer_amd import group as lm_group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        limit = int(os.environ.get("LUMEN_BENCH_RCCL_TIMEOUT_S", "300"))
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit))
        state = {"stage": "communicator id"}

        def watchdog():                                   # RCCL has no timeout of its own: a rank that cannot reach its peers says so and ends the run instead of hanging it
            sys.stderr.write(f"bench.py rank {rank} of {world} ({my_dev_early}): native tile group stuck in '{state['stage']}' for {limit} s — peers unreachable? "
                             f"(HSA_ENABLE_IPC_MODE_LEGACY=0 exported? every rank on its own GPU? LUMEN_BENCH_TRANSPORT=torch selects the torch.distributed path)\n")
            sys.stderr.flush()
            os._exit(3)
        my_dev_early = f"cuda:{local_rank}"
        timer = threading.Timer(limit, watchdog); timer.daemon = True; timer.start()
        if one_gpu:
            transport, gid = lm_group.DistHostTransport(dist), None
        else:
            box = [lm_group.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            transport, gid = None, box[0]
        state["stage"] = "communicator creation"
        native_error = ""
        try:
            if os.environ.get("LUMEN_BENCH_NATIVE_FAIL"):    # test hook: the fallback below, exercised where the native group would work
                raise RuntimeError("forced by LUMEN_BENCH_NATIVE_FAIL")
            grp = lm_group.TileGroup(r, rank, world, group_id=gid, transport=transport)
            assert grp.tile == tuple(tile) and grp.window == tuple(win), (grp.tile, tile, grp.window, win)
            state["stage"] = "self-test (all-reduce on both communicators + a full-size gather)"
            self_test_ms = grp.SelfTest()
        except Exception as ex:                              # e.g. librccl not resolvable, communicator creation refused: every rank sees the same and says so below
            native_error = f"{type(ex).__name__}: {ex}"
        timer.cancel()
        # the ranks agree (gloo) whether the native group stands; if it does not anywhere, ALL of them fall back to the torch.distributed transport and the line says so
        flag = torch.tensor([1 if native_error else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            sys.stderr.write(f"bench.py rank {rank}: native tile group unavailable ({native_error or 'failed on another rank'}); falling back to --transport torch\n"); sys.stderr.flush()
            if grp is not None:
                try:
                    grp.close()
                except Exception:
                    pass
            grp, native = None, False
            args.transport = "torch (fallback: native tile group unavailable" + (": " + native_error[:160] if native_error else "") + ")"
            dist.destroy_process_group()
    if (world > 1 or force_pg) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from host_staged_dist import HostStagedDist          # test infrastructure (tests/): gloo through host staging, rehearsal only
            dist = HostStagedDist(dist)
        else:
            import datetime
            try:                                                                                     # an explicit timeout: a rank that never arrives ends the run with a message, not a hang
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("LUMEN_BENCH_RCCL_TIMEOUT_S", "300"))))
            except Exception as ex:
                raise SystemExit(f"bench.py rank {rank}: the RCCL communicator over {world} GPUs could not be created ({type(ex).__name__}: {ex}); "
                                 f"{present} GPU(s) visible here — is HSA_ENABLE_IPC_MODE_LEGACY=0 exported and does every rank see its device?")
        warm = torch.zeros(1, device=dev); dist.all_reduce(warm)                                   # communicator + stream exist now
        torch.cuda.synchronize()

    # temporal history across tile seams: only a path depth with an odd number of waves per frame has any (tiles.history_needed);
    # then every TraceFrame is followed by one grouped RCCL send / recv of the halo rings' reservoirs
    exact = args.exact_seams == "on" or (args.exact_seams == "auto" and tiles.history_needed(depth))
    hx = tiles.HistoryExchange(r, rank, world, W, H, dev) if (world > 1 and exact and not native) else None

    ev_log = []                                           # per step: torch events around render / seam exchange / gather (multi-GPU explainers)

    pose_no = [0]

    def frame(record=False):
        r.SetBlendMode(blend_on)                          # a fresh 4-spp accumulation per displayed frame (a moving camera: blending off, OutputLayer.cpp:492-495)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record:
            evs[0].record()
        for _ in range(spp):
            if moving:
                pose_no[0] += 1
                r.SetCamera(*camera_pose(desc, pose_no[0]))
            if grp is not None and exact:
                grp.TraceFrame()                          # TraceFrame + (odd depths) wave-count agreement + ONE grouped seam exchange, in C++
            else:
                r.TraceFrameAsync()
            if hx is not None:
                hx.run(dist)
        if record:
            evs[1].record()
        # the tile (halo removed) goes from the renderer's merged radiance straight into the gather's send buffer, the gathered tiles into the frame on rank 0: the
        # library's own pitched-copy kernel on the renderer's stream (tiles.TileGather.run_renderer), then ONE gather (RCCL over xGMI)
        if emu:
            r.CopyRadianceRectToDevice(tile, emu_tile_buf.data_ptr(), tile[2] - tile[0]); out = emu_tile_buf
        elif grp is not None:
            grp.Gather(); out = None                      # enqueue only: tile -> send tile -> rank 0 on the gather stream, overlapping the next step's rendering
        else:
            out = tiles.gather_from_renderer(r, rank, world, W, H, dist, dev)
        if record:
            evs[2].record(); ev_log.append(evs)
        return out

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented pass (outside the timed region): BVH nodes / triangles per closest-hit ray for the roofline
    r.SetInstrumented(True)
    r.SetBlendMode(blend_on)
    r.TraceFrame()
    ci = r.GetCounters(50)
    # 64-byte node records fetched: lane-level node steps of the queue traversal kernels (counter 41); child boxes tested / 4 where the
    # counting build ran the path tail instead (its steps are not in counter 41)
    node_records = float(ci[41]) if ci[41] > 0 else ci[22] / 4.0
    r.SetInstrumented(False)

    def timed_pass(fast, lazy):
        """W warm-up steps, then exactly K steps between barriers; returns the per-rank measurements of the pass."""
        r.SetTuning("fast_resample", 1 if fast else 0)
        r.SetTuning("lazy_reuse", -1 if lazy else 0)
        del ev_log[:]
        for _ in range(args.warmup):
            frame()
        # HIP events around the roofline kernel's launches (and one pair per TraceFrame) only; the per-class breakdown comes from the pass below
        r.EnableKernelTiming(0 if args.no_kernel_timing else 2)
        barrier()
        r.GetCounterTotals(4, reset=True)                 # device-side running sums of the ray counters: from here on, the timed TraceFrames only
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame(record=world > 1)
        host_dt = time.perf_counter() - t0                # the host has queued every launch of the K steps; the device is still working
        barrier()
        dt = time.perf_counter() - t0
        r.EnableKernelTiming(False)
        # counters of the LAST TraceFrame (every TraceFrame of a step traces the same number of rays to within RNG noise); reading them also
        # collects the HIP-event times of the timed region, on the stream the kernels were launched on
        c = r.GetCounters(50)
        ct = r.GetCounterTotals(50)                       # summed over all K x spp timed TraceFrames
        assert ct[3] == args.steps * spp, (ct[3], args.steps, spp)
        k = {name: r.GetKernelTime(i) for i, name in enumerate(("closest", "shadow", "shade", "restir", "total", "tail"))}
        # device time by kernel class (explainer, outside the timed region): two more steps with events around every launch
        r.EnableKernelTiming(1)
        for _ in range(2):
            frame()
        barrier()
        r.EnableKernelTiming(False)
        r.GetCounters(50)
        kb = {name: r.GetKernelTime(i) for i, name in enumerate(("closest", "shadow", "shade", "restir", "total", "tail"))}
        n_tf = max(1, k["total"][1])
        # a rank's counters include the rays of its halo pixels; those are redundant work (the neighbour owns the pixels), so only the
        # tile's share is counted: primary rays and the first ReSTIR visibility pass cover the whole window (scaled to the tile);
        # indirect waves, NEE and the second visibility pass only run for tile pixels already (lumen_mi_set_tile)
        tile_share = ((tile[2] - tile[0]) * (tile[3] - tile[1])) / float(ww * wh)
        rays_tile = (ct[4] + ct[48]) * tile_share + (ct[0] - ct[4]) + ct[1] + ct[49] if (world > 1 or emu) else float(ct[0] + ct[1] + ct[2])      # all timed steps
        render_ms = sum(e[0].elapsed_time(e[1]) for e in ev_log) / max(1, len(ev_log)) if ev_log else None
        gather_ms = sum(e[1].elapsed_time(e[2]) for e in ev_log) / max(1, len(ev_log)) if ev_log else None
        if grp is not None:
            gather_ms = grp.Stats()["mean_gather_ms"]     # HIP events on the gather stream (transport + placement), not a stall of the render stream
        stats = torch.tensor([dt, float(rays_tile), render_ms or 0.0, gather_ms or 0.0], dtype=torch.float64, device="cpu" if grp is not None else dev)
        per_rank = None
        if world > 1:
            allst = [torch.zeros_like(stats) for _ in range(world)]
            dist.all_gather(allst, stats)
            dt = max(float(t[0]) for t in allst); rays_all = sum(float(t[1]) for t in allst)
            halo = lambda i: (lambda t_, w_: round(((w_[2] - w_[0]) * (w_[3] - w_[1])) / float((t_[2] - t_[0]) * (t_[3] - t_[1])) - 1.0, 4))(tiles.tile_rect(i, world, W, H), tiles.window_rect(tiles.tile_rect(i, world, W, H), W, H))
            per_rank = [{"rank": i, "wall_ms_per_step": round(float(t[0]) * 1e3 / args.steps, 3), "render_ms_per_step": round(float(t[2]), 3),
                         "gather_ms_per_step": round(float(t[3]), 3), "gather_ms": round(float(t[3]), 3), "halo_over_tile": halo(i)} for i, t in enumerate(allst)]
        else:
            rays_all = float(rays_tile)
        ms_per_step = dt * 1e3 / args.steps
        return {"dt": dt, "ms_per_step": ms_per_step, "host_submit_ms_per_step": host_dt * 1e3 / args.steps, "rays_per_frame": rays_all / args.steps, "value": rays_all / dt / 1e6,
                "counters": c, "kernel_ms": k, "class_ms": kb, "n_traceframes": n_tf, "per_rank": per_rank}

    # who took part: the size of the RCCL communicator the gather ran on and every rank's device, so that a scaling record shows its N ranks by itself
    my_dev = f"{torch.cuda.get_device_name(local_rank)} (cuda:{local_rank})"
    rccl_world = world if grp is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    devices = [my_dev]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, my_dev)

    fast = args.mode == "fast"
    lazy = args.reuse == "lazy"
    main_pass = timed_pass(fast, lazy)
    other_pass = None if (args.no_exact or world > 1 or emu) else timed_pass(not fast, lazy)
    # the other setting of the history passes, both arithmetic modes (single GPU only: the extra passes would double a scaling run)
    # (an odd path depth has no lazy reuse — the swap chain turns every frame and the history is read — so the two settings are the same run: skipped)
    odd = depth % 2 == 1
    reuse_pass = None if (args.no_other_reuse or world > 1 or emu or odd) else timed_pass(fast, not lazy)
    reuse_other = None if (args.no_other_reuse or args.no_exact or world > 1 or emu or odd) else timed_pass(not fast, not lazy)
    # the same frame with FULL persistent traversal grids everywhere (what rounds 1 - 5 ran): the roofline kernel's launch is faster that way and the frame slower; both on the line
    full_grid_pass = None
    if not (args.no_other_reuse or world > 1 or emu):
        r.SetTuning("trace_blocks_main", 8); r.SetTuning("trace_blocks_vis", 8)
        full_grid_pass = timed_pass(fast, lazy)
        r.SetTuning("trace_blocks_main", 0); r.SetTuning("trace_blocks_vis", 0)
    r.SetTuning("fast_resample", 1 if fast else 0); r.SetTuning("lazy_reuse", -1 if lazy else 0)
    passes = {(fast, lazy): main_pass, (not fast, lazy): other_pass, (fast, not lazy): reuse_pass, (not fast, not lazy): reuse_other}      # (fast?, lazy?) -> pass or None
    rate = lambda f, l, key="value", nd=3: None if passes[(f, l)] is None else round(passes[(f, l)][key], nd)
    if rank == 0:
        c, k, n_tf = main_pass["counters"], main_pass["kernel_ms"], main_pass["n_traceframes"]
        ms_per_step, value, rays_per_frame = main_pass["ms_per_step"], main_pass["value"], main_pass["rays_per_frame"]
        all_rays_inst = max(1, ci[0] + ci[1] + ci[2])
        # ---- roofline of the dominant kernel by device time (closest-hit traversal: one launch per wave in front of the path tail)
        closest_ms, closest_launches = k["closest"]
        # rays the closest-hit KERNELS traced: the waves before the path tail takes over (the tail is its own launch and timing class)
        rays_ck = float(sum(c[4 + d] for d in range(min(depth, int(round(closest_launches / n_tf))))))
        nodes4_c = node_records * (rays_ck / all_rays_inst)                 # 64-byte node records fetched, closest-hit kernels' share
        tris_c = ci[21] * (rays_ck / all_rays_inst)
        alg = algorithmic_bytes_closest(rays_ck, nodes4_c, tris_c)          # per TraceFrame (all closest-hit launches)
        alg_d4 = algorithmic_bytes_closest(rays_ck, 2.0 * nodes4_c, tris_c) # the same visits priced as binary nodes (SURVEY d4 wording)
        launches_per_tf = closest_launches / n_tf
        per_launch_ms = closest_ms / max(1, closest_launches)
        gbs = lambda bytes_per_tf: (bytes_per_tf / max(1.0, launches_per_tf)) / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        achieved = gbs(alg)
        pmc = load_pmc() if args.workload == "c2" and world == 1 and not emu else {}
        pk = lambda name, key: pmc.get(name, {}).get(key)
        # per launch, averaged over both closest-hit kernels (per-lane queue kernel + the packet kernel of the primary wave)
        # HBM bytes per launch from the PMC replay: FETCH_SIZE x 2 for streaming kernels, x 1 for gather kernels (tools/pmc_json.sh: field `access`), + WRITE_SIZE;
        # a replay file of rounds 2 - 5 only has the all-x2 upper bound (`hbm_bytes_per_launch_corrected`), which stays beside it as *_upper
        hb = lambda v: v.get("hbm_bytes_per_launch", v.get("hbm_bytes_per_launch_corrected", 0.0))
        tc = [(hb(pmc.get(n, {})) or None, pk(n, "launches")) for n in ("lm_k_trace_closest", "lm_k_trace_closest_packet")]
        tc = [(b, l) for b, l in tc if b is not None and l]
        traffic_closest = sum(b * l for b, l in tc) / sum(l for _, l in tc) if tc else None
        # the same algorithmic bytes over the ALONE launch time (PMC passes serialise the dispatches): what the kernels reach when nothing else is resident
        ta = [(pk(n, "alone_us"), pk(n, "launches")) for n in ("lm_k_trace_closest", "lm_k_trace_closest_packet")]
        ta = [(u, l) for u, l in ta if u and l]
        alone_us = sum(u * l for u, l in ta) / sum(l for _, l in ta) if ta else None
        frac_alone = None if not alone_us else round((alg / max(1.0, launches_per_tf)) / (alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        # PHYSICAL HBM fraction of every kernel that moves more than 50 MB per launch: PMC bytes / alone time / peak (which passes are byte-bound, which are not)
        hbm_kernels = sorted(({"kernel": n, "access": v.get("access"), "hbm_mb_per_launch": round(hb(v) / 1e6, 1), "hbm_mb_per_launch_upper": round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1), "alone_us": round(v["alone_us"], 1),
                               "frac": round(hb(v) / (v["alone_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
                               "active_lanes_per_valu_inst": round(v.get("active_lanes_per_valu_inst") or 0.0, 1)}
                              for n, v in pmc.items() if v.get("alone_us") and v.get("hbm_bytes_per_launch_corrected", 0.0) > 50e6 and v.get("launches", 0) > 2),
                             key=lambda e: -e["frac"])
        npix = (win[2] - win[0]) * (win[3] - win[1])
        alg_tf = algorithmic_bytes_traceframe(c, depth, npix, ci[20], ci[21], blend=blend_on)
        # whole-frame HBM traffic from the PMC replay: sum over kernels of bytes per launch x launches per TraceFrame
        tf_in_pmc = max(1, pmc.get("lm_k_primary", {}).get("launches", 1))
        hbm_tf = sum(hb(v) * v.get("launches", 0) / tf_in_pmc for v in pmc.values()) if pmc else None
        hbm_tf_upper = sum(v.get("hbm_bytes_per_launch_corrected", 0.0) * v.get("launches", 0) / tf_in_pmc for v in pmc.values()) if pmc else None
        tf_ms = ms_per_step / spp
        # ---- VALU-bound kernels (candidate pick; spatial reuse): executed lane-operations / alone time against the fp32 issue peak
        valu = []
        pick = "lm_k_pick_primary_fast" if fast else "lm_k_pick_primary"
        ns = lambda n: n + "_ns" if n + "_ns" in pmc else n              # kernels that run from the compilation without the SLP vectoriser carry the suffix (renderer.cpp applyNoSlpKernels)
        pick = ns(pick + "_lds") if ns(pick + "_lds") in pmc else ns(pick)           # scenes whose light table fits in LDS run that instantiation
        for name in (pick, ns("lm_k_restir_spatial_fast" if fast else "lm_k_restir_spatial")):
            lane_ops, us = pk(name, "SQ_THREAD_CYCLES_VALU_per_launch"), pk(name, "alone_us")
            if lane_ops and us:
                ach = lane_ops / (us * 1e-6) / 1e12
                valu.append({"kernel": name, "bound": "valu", "achieved": round(ach, 2), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                             "frac": round(ach / VALU_PEAK_TLANEOPS, 4), "alone_us": round(us, 1), "active_lanes_per_inst": round(pk(name, "active_lanes_per_valu_inst") or 0.0, 1),
                             "valu_insts_per_launch": int(pk(name, "SQ_INSTS_VALU_per_launch") or 0), "source": PMC_FILE + " (replayed)"})
        # ---- what binds the frame, and what binds the traversal class
        # frame level: wave-instructions the vector ALUs issued per TraceFrame (PMC replay, all kernels) against what the chip issues with dependent chains at 8 waves
        # per SIMD (1 055.8 G wave-instructions/s measured, profiles/r03_valu_peak.txt), beside the HBM fraction: neither is near 1 — the frame is latency-bound
        frame_valu = sum(v.get("SQ_INSTS_VALU_per_launch", 0.0) * v.get("launches", 0) / tf_in_pmc for v in pmc.values()) if pmc else None
        frame_valu_frac = None if not frame_valu else round(frame_valu / (tf_ms * 1e-3) / WAVEINST_PEAK, 4)
        frame_hbm_frac = None if hbm_tf is None else round(hbm_tf / (tf_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
        # traversal class: a launch of incoherent rays takes T0 + rays / R — T0 the dependent chain of its longest ray, R the VALU issue rate at the lane occupancy
        # divergence leaves (both measured on this scene: profiles/r04_step_latency.txt).  `issue_bound_ms` = the launch's own wave-instructions at the issue peak.
        queue_launches = max(0, int(round(launches_per_tf)) - 1)            # the primary wave is the packet kernel; waves 1.. are the queue kernel
        rays_queue = float(sum(c[4 + d] for d in range(1, 1 + queue_launches)))
        steps_mean = (ci[22] / 4.0 + ci[21]) / all_rays_inst                # traversal steps per ray of this frame (4-wide node steps + triangle tests, counting build)
        rays_per_us = TRAV_STEPS_PER_US / max(1.0, steps_mean)
        tm_pred_ms = (queue_launches * TRAV_T0_US + rays_queue / rays_per_us) * 1e-3 if queue_launches else None
        v_tc = pk("lm_k_trace_closest", "SQ_INSTS_VALU_per_launch")
        traversal_model = None if not queue_launches else {
            "kernel": "lm_k_trace_closest (waves 1.." + str(queue_launches) + ")", "launches_per_traceframe": queue_launches, "rays_per_traceframe": int(rays_queue),
            "model": "T = T0 + rays x steps_per_ray / S per launch; T0 = dependent chain of the longest ray, S = traversal steps per microsecond the chip sustains at the lane occupancy divergence leaves (4.6 G rays/s x 16.3 steps measured on incoherent rays, profiles/r04_step_latency.txt)",
            "t0_us": TRAV_T0_US, "steps_per_us": TRAV_STEPS_PER_US, "rays_per_us": round(rays_per_us, 1), "predicted_ms_per_traceframe_alone": round(tm_pred_ms, 4),
            "issue_bound_ms_per_traceframe": None if not v_tc else round(queue_launches * v_tc / WAVEINST_PEAK * 1e3, 4),
            "alone_ms_per_traceframe": None if not pk("lm_k_trace_closest", "alone_us") else round(queue_launches * pk("lm_k_trace_closest", "alone_us") * 1e-3, 4),
            "active_lanes_per_valu_inst": pk("lm_k_trace_closest", "active_lanes_per_valu_inst"),
            "steps_per_ray_mean": round(steps_mean, 2), "steps_longest_ray": int(ci[40]),
            "note": "alone ~ predicted means the launches sit ON the chain + issue bound; the live launch_ms of `roofline` is longer because three other streams share the machine"}
        dev = lambda kk: {n: round(kk[n][0] / max(1, kk["total"][1]), 3) for n in kk}
        out = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "rccl_world": rccl_world, "devices": devices,
            **({"transport": ("native tile group (csrc/group.cpp): " + ("host transport over gloo, rehearsal" if one_gpu else "RCCL resolved by the library, two communicators, gather on its own stream, double-buffered")) if grp is not None else ("torch.distributed (tiles.py)" + (" — " + args.transport if str(args.transport).startswith("torch (fallback") else "")),
                "group_self_test_ms": round(self_test_ms, 2) if grp is not None else None} if world > 1 else {}),
            **({"rehearsal": f"{world} ranks share ONE GPU over gloo with host staging (LUMEN_BENCH_ONE_GPU=1): the N-rank code path executed, NOT a measurement"} if one_gpu else {}),
            # both arithmetic modes at the top level, so that `value` cannot be read without its mode: "fast" = hardware rcp / rsq / sqrt +
            # contracted target function in the ReSTIR passes (the reference's Release build is -use_fast_math); "exact" = bit-identical to the oracle
            "mode": args.mode, "value_fast": round((main_pass if fast else other_pass)["value"], 3) if (fast or other_pass) else None,
            "value_exact": round((other_pass if fast else main_pass)["value"], 3) if (not fast or other_pass) else None,
            "ms_per_step_exact": round((other_pass if fast else main_pass)["ms_per_step"], 4) if (not fast or other_pass) else None,
            # history passes (see the module docstring): "eager" = both spatial passes + combine launched in every TraceFrame like the reference's; "lazy" = the
            # renderer's default, launched when their result can be read.  Identical images, counters and rays; both rates in the mode of `value`
            "reuse": args.reuse,
            "value_eager_reuse": rate(fast, False), "value_lazy_reuse": rate(fast, True), "ms_per_step_lazy_reuse": rate(fast, True, "ms_per_step", 4),
            "value_exact_lazy_reuse": rate(False, True), "value_exact_eager_reuse": rate(False, False),
            "config": {"workload": f"{args.workload}: {'cornell box (the reference asset, tests/golden fixture)' if kind == 'cornell' else 'LowpolyRoom/scene.glb (the Sandbox default model, tests/golden fixture)' if kind == 'lowpoly' else kind + ' stand-in'}, {W}x{H}, " +
                                   ("1 TraceFrame per step at a new camera pose (blending off: the Sandbox's own default setting, Application.cpp:89-93, live temporal history)" if moving else f"{spp} spp (blended TraceFrames)") +
                                   f", depth {depth}, ReSTIR DI on",
                       "resample_mode": ("fast: hardware rcp/rsq/sqrt + contracted target function in the ReSTIR passes (rel-L2 vs oracle 1e-8 measured, 1e-3 asserted: "
                                         "test_fast_resampling_mode_stays_within_the_north_star_tolerance)") if fast else "exact: correctly rounded everywhere, bit-identical to the oracle",
                       "other_mode": None if other_pass is None else {"mode": "exact" if fast else "fast", "value": round(other_pass["value"], 3), "ms_per_step": round(other_pass["ms_per_step"], 4),
                                                                     "device_ms_per_traceframe": dev(other_pass["class_ms"])},
                       "history_passes": ("eager: both spatial reuse passes and the reservoir combine are launched in every TraceFrame, as the reference launches them (tuning key lazy_reuse 0)"
                                          if not lazy else "lazy (the renderer's default at even path depths): both spatial reuse passes and the reservoir combine run when their result can be read") +
                                         "; they only build the next frame's history, which the reference's swap quirk never reads at an even path depth — images, counters and rays are identical "
                                         "either way (test_history_passes_run_only_when_their_result_can_be_read), both rates are on this line: value_eager_reuse / value_lazy_reuse",
                       "triangles": desc.triangle_count(), "rays_per_frame": int(rays_per_frame), "ms_per_frame": round(ms_per_step, 4),
                       "tiles": f"{tiles.grid_for(world, W, H)[0]}x{tiles.grid_for(world, W, H)[1]} + {tiles.HALO}px halo" if world > 1 else "1x1",
                       "nodes4_per_ray": round(node_records / all_rays_inst, 2), "binary_node_equivalents_per_ray": round(ci[20] / all_rays_inst, 2), "tris_per_ray": round(ci[21] / all_rays_inst, 2),
                       "rays_per_wave": [int(c[4 + d]) for d in range(depth)], "nee_shadow_rays": int(c[1]), "restir_shadow_rays": int(c[2]),
                       "kernel_mix": "kernels from the compilation without the SLP vectoriser: " + os.environ.get("LUMEN_MI_NOSLP_KERNELS", "pick_primary,extract0,shade_wave,merge") +
                                     " (+ in the exact mode: " + os.environ.get("LUMEN_MI_NOSLP_KERNELS_EXACT", "temporal,spatial,combine") + "); profiles/r06_noslp_kernels_ab.txt",
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "host_submit_ms_per_step": round(main_pass["host_submit_ms_per_step"], 3),
                       "d4_accounting_bytes_per_traceframe": int(alg_tf),
                       "d4_accounting_over_hbm_peak": round(alg_tf * spp / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) if world == 1 else None,
                       "d4_note": "SURVEY d4 prices the reference's AoS data flow; above 1.0 means most of those bytes are cache hits or never move here — it is not a roofline",
                       "hbm_traffic_bytes_per_traceframe": None if hbm_tf is None else int(hbm_tf),
                       "hbm_traffic_frac": None if hbm_tf is None else round(hbm_tf / (tf_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                       "hbm_traffic_source": None if hbm_tf is None else PMC_FILE + " (replayed: PMC passes of the builder's run; FETCH_SIZE x 2 for streaming kernels, x 1 for gather kernels, + WRITE_SIZE; frame_hbm_frac_upper = x 2 everywhere)"},
            # `bound`: the kernel is NOT HBM-bound — the tree is served by L2 / Infinity Cache (frac_hbm_physical: a few % of peak); what bounds a launch is the dependent
            # chain of its longest rays x the lanes divergence leaves idle (traversal_model).  `frac` stays the contract's figure (SURVEY d4 algorithmic bytes / live
            # launch time / HBM peak) so that rounds compare; `frac_alone` is the same bytes over the kernels' serialised (alone) time from the PMC replay.
            "roofline": {"bound": "latency/divergence", "kernel": "lm_k_trace_closest (+ lm_k_trace_closest_packet: the primary wave)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_alone": frac_alone, "traffic": traffic_closest,
                         # the PHYSICAL fraction beside the algorithmic one: PMC bytes per launch / live launch time / peak
                         "frac_hbm_physical": None if (not traffic_closest or per_launch_ms <= 0) else round(traffic_closest / (per_launch_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 5),
                         "traffic_source": None if traffic_closest is None else PMC_FILE + " (replayed, not measured by this run)",
                         "traffic_over_algorithmic": None if not traffic_closest else round(traffic_closest / (alg / max(1.0, launches_per_tf)), 4),
                         "limiter": "dependent-load latency x lane divergence: the tree is served by L2 / Infinity Cache, HBM sees a few % of the algorithmic bytes",
                         "grid_note": "since round 6 the primary-ray launch of an eager frame runs on HALF a persistent grid on purpose (csrc/frame.cpp; profiles/r06_trace_blocks_ab.txt): that launch "
                                      "got slower (611 -> 730 us under overlap) and the frame faster, so frac is below the 0.545 of the full-grid record at a higher value; frac_full_grid / value_full_grid = the same run "
                                      "with tuning keys trace_blocks_main = trace_blocks_vis = 8",
                         **({} if full_grid_pass is None or full_grid_pass["kernel_ms"]["closest"][1] == 0 else
                            {"frac_full_grid": round(achieved / HBM_PEAK_GBS * per_launch_ms / (full_grid_pass["kernel_ms"]["closest"][0] / full_grid_pass["kernel_ms"]["closest"][1]), 5),
                             "value_full_grid": round(full_grid_pass["value"], 3)}),
                         "launch_ms": round(per_launch_ms, 4), "launches_per_traceframe": launches_per_tf,
                         "algorithmic_bytes_per_launch": int(alg / max(1.0, launches_per_tf)),
                         "achieved_d4_binary_node_pricing": round(gbs(alg_d4), 2)},
            "roofline_valu": valu,
            "roofline_hbm_kernels": hbm_kernels,
            "frame_valu_frac": frame_valu_frac, "frame_hbm_frac": frame_hbm_frac,
            "frame_hbm_frac_upper": None if not pmc or hbm_tf_upper is None else round(hbm_tf_upper / (tf_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
            "frame_bound_note": "fractions of the chip's VALU issue peak (wave-instructions, measured) and of HBM peak over one TraceFrame, from the PMC replay: both well below 1 — "
                                "the frame is bound by dependent-chain latency inside the traversal launches and by how well four streams fill each other's stalls",
            "traversal_model": traversal_model,
            "device_ms_per_traceframe": dev(main_pass["class_ms"]),
        }
        if main_pass["per_rank"]:
            out["per_rank"] = main_pass["per_rank"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, kw, depth, spp, (W, H), moving)
        if emu:
            out["emulated_rank"] = {"rank": emu[0], "of": emu[1], "window": list(win), "tile": list(tile),
                                    "note": "per-rank time of the tiled path on one GPU; value counts the tile's rays only"}
        try:                                   # RCCL writes a version banner through C stdio: flush it first so that the JSON is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if grp is not None:
        if rank == 0:
            img = grp.GetFrame()                          # the last gathered frame exists and holds light
            assert float(img[..., :3].sum()) > 0.0
        grp.close()
    r.close()
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
