#!/usr/bin/env python3
"""usage (GPU box): [LUMEN_MI_LIBRARY=<build>] python tools/scene_ms.py <scene.npz> [W H depth] — ms per TraceFrame and Mrays/s of a scene file (lumenrenderer_amd.scenes.scene_to_npz with
textures) in fast and exact mode, eager reuse, 32 asynchronous frames after 8 warm-up frames, median of 3; + the device time by kernel class.  For looking at content beyond bench.py's workloads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import product_from
from lumenrenderer_amd.scenes import scene_from_npz
path = sys.argv[1]; W, H, D = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (1280, 720, 5)
d = scene_from_npz(path)
for mode in (1, 0):
    r = product_from(d, W, H, D, blend=False, tuning={"fast_resample": mode, "lazy_reuse": 0})
    for _ in range(8): r.TraceFrameAsync()
    r.Synchronize()
    ts = []
    for _ in range(3):
        r.GetCounterTotals(4, reset=True)
        t0 = time.perf_counter()
        for _ in range(32): r.TraceFrameAsync()
        r.Synchronize()
        dt = time.perf_counter() - t0
        ct = r.GetCounterTotals(8)
        ts.append((dt / 32 * 1e3, (ct[0] + ct[1] + ct[2]) / dt / 1e6))
    ts.sort()
    r.EnableKernelTiming(1)
    for _ in range(4): r.TraceFrameAsync()
    r.Synchronize(); r.EnableKernelTiming(False); c = r.GetCounters(12)
    kb = {n: round(r.GetKernelTime(i)[0] / max(1, r.GetKernelTime(4)[1]), 3) for i, n in enumerate(("closest", "shadow", "shade", "restir", "total", "tail"))}
    print(f"{os.path.basename(os.environ.get('LUMEN_MI_LIBRARY', 'default'))} {os.path.basename(path)} {W}x{H} depth {D} {'fast' if mode else 'exact'}: {ts[1][0]:.3f} ms per TraceFrame, {ts[1][1]:.0f} Mrays/s; lights {c[3]}, rays per wave {c[4:4 + D]}; device ms by class {kb}")
    r.close()
