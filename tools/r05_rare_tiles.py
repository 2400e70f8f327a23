#!/usr/bin/env python3
"""usage (GPU box): LUMEN_MI_LIBRARY=<build> python tools/r05_rare_tiles.py — what ONE small glass object costs the fast ReSTIR mode: the stand-in atrium at 1440p, depth 6, with the 960
triangles of material 22 turned into glass (transmission 0.6), against the unchanged scene; ms per TraceFrame (fast mode, eager reuse, 24 asynchronous frames after 8 warm-up frames, 3 repeats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import product_from
from lumenrenderer_amd import scenes


def ms_per_frame(desc, W=2560, H=1440, D=6):
    r = product_from(desc, W, H, D, blend=True, tuning={"fast_resample": 1, "lazy_reuse": 0})
    for _ in range(8): r.TraceFrameAsync()
    r.Synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(24): r.TraceFrameAsync()
        r.Synchronize()
        out.append((time.perf_counter() - t0) / 24 * 1e3)
    g = r.GetGBuffer(); p2 = g[..., 7, 2].copy().view(np.uint32); flags = g[..., 1, 3].copy().view(np.uint32)
    share = float(((flags == 0) & (((p2 >> 16) & 0xff) != 0)).mean())
    r.close()
    return sorted(out)[1], share


plain = scenes.sponza_standin()
glass = scenes.sponza_standin(); glass.materials[22]["transmission_factor"] = 0.6
a, sa = ms_per_frame(plain); b, sb = ms_per_frame(glass)
print(f"{os.path.basename(os.environ.get('LUMEN_MI_LIBRARY', 'default build'))}: plain {a:.3f} ms per TraceFrame (glass surfaces {sa:.5f} of the pixels) | one glass object {b:.3f} ms ({sb:.5f} of the pixels): {(b / a - 1) * 100:+.1f} %")
