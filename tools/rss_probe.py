"""Diagnostic (GPU box): CURRENT host RSS (not the high-water mark) and device memory every 2000 frames of the soak edit sequence,
with the host free to run ahead of the device (sync_every = 0) or synchronised every N frames.  python tools/rss_probe.py <frames> <sync_every>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy, time
import numpy as np
import torch
N, SYNC = int(sys.argv[1]), int(sys.argv[2])
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak.py")).read().split("r = LumenRendererMI()")[0]
sys.argv = [sys.argv[0], str(N)]
exec(src)                                           # imports, scene, edits(), used()
def rss(): return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20
r = LumenRendererMI(); r.Init(depth=6, render_resolution=(W, H), blend_output=False)
r.LoadSceneDescription(desc)
for k in range(50): r.TraceFrameAsync()
r.Synchronize()
t0 = time.perf_counter()
for k in range(N):
    edits(r, k)
    assert r.TraceFrameAsync()
    if SYNC and k % SYNC == SYNC - 1: r.Synchronize()
    if k % 2000 == 1999:
        ahead = time.perf_counter() - t0
        print(f"frame {k + 1}: host at {ahead:.1f} s, rss {rss():.0f} MiB, device {used():.0f} MiB", flush=True)
r.Synchronize()
print(f"done in {time.perf_counter() - t0:.1f} s: rss {rss():.0f} MiB, device {used():.0f} MiB")
r.close()
