"""Upper bound for deeper frame pipelining (GPU box): two independent renderers on one GPU, frames enqueued alternately, against one
renderer alone.  If two together are not clearly faster than one, more frames in flight cannot help either.  python tools/two_renderers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import time
from lumenrenderer_amd import LumenRendererMI, tiles
from lumenrenderer_amd.scenes import sponza_standin
W, H, D = 2560, 1440, 6
desc = sponza_standin()


def make(n_ranks):
    r = LumenRendererMI(); r.Init(depth=D, render_resolution=(W, H), blend_output=True)
    r.LoadSceneDescription(desc)
    if n_ranks > 1:
        t = tiles.tile_rect(1, n_ranks, W, H); r.SetWindow(*tiles.window_rect(t, W, H)); r.SetTile(*t)
    return r


for n_ranks in (1, 4, 8):
    for count in (1, 2):
        rs = [make(n_ranks) for _ in range(count)]
        for _ in range(8):
            for r in rs: r.TraceFrameAsync()
        for r in rs: r.Synchronize()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            for r in rs: r.TraceFrameAsync()
        for r in rs: r.Synchronize()
        dt = time.perf_counter() - t0
        print(f"window of 1/{n_ranks}: {count} renderer(s): {dt / (n * count) * 1e3:.3f} ms per TraceFrame")
        for r in rs: r.close()
