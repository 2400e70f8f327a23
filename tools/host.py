"""Host enqueue cost of TraceFrameAsync vs device time, full frame and the window of one rank of 8 (GPU box): python tools/host.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import time, numpy as np
from lumenrenderer_amd import LumenRendererMI, tiles
from lumenrenderer_amd.scenes import sponza_standin
W, H = 2560, 1440
for n_ranks in (1, 8):
    r = LumenRendererMI(); r.Init(depth=6, render_resolution=(W, H), blend_output=True)
    r.LoadSceneDescription(sponza_standin())
    if n_ranks > 1:
        tile = tiles.tile_rect(1, n_ranks, W, H); r.SetWindow(*tiles.window_rect(tile, W, H)); r.SetTile(*tile)
    for _ in range(8): r.TraceFrameAsync()
    r.Synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): r.TraceFrameAsync()
    t1 = time.perf_counter()
    r.Synchronize()
    t2 = time.perf_counter()
    print(f"ranks={n_ranks}: host enqueue {(t1 - t0) / n * 1e3:.3f} ms per TraceFrame, total {(t2 - t0) / n * 1e3:.3f} ms per TraceFrame")
    r.close()
