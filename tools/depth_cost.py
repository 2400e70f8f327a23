"""What the deep waves cost the frame (GPU box): C2's window and scene at path depth 6 / 4 / 2 (even: the history passes behave alike), default schedule and with
the path tail off (tail_below 0: every wave its own launches on the wave stream).  Prints ms per TraceFrame, eager fast mode.  python tools/depth_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lumenrenderer_amd import LumenRendererMI
from lumenrenderer_amd.scenes import sponza_standin

desc = sponza_standin()


def run(depth, tuning, frames=40, warm=8):
    r = LumenRendererMI(); r.Init(depth=depth, render_resolution=(2560, 1440), blend_output=True)
    r.LoadSceneDescription(desc); r.SetBlendMode(True)
    for k, v in dict(fast_resample=1, lazy_reuse=0, **tuning).items(): r.SetTuning(k, v)
    for _ in range(warm): assert r.TraceFrameAsync()
    r.Synchronize()
    t0 = time.perf_counter()
    for _ in range(frames): assert r.TraceFrameAsync()
    r.Synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    c = r.GetCounters()
    r.close()
    return ms, [int(x) for x in c[4:4 + depth]]


for rnd in range(2):
    for depth, tuning in ((6, {}), (6, {"tail_below": 0}), (4, {}), (4, {"tail_below": 0}), (2, {}), (6, {"tail_below": 300000})):
        ms, rays = run(depth, tuning)
        print(f"depth {depth} {str(tuning):24s} {ms:7.3f} ms per TraceFrame   rays per wave {rays}", flush=True)
