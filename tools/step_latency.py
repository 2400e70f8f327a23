#!/usr/bin/env python3
"""What bounds a closest-hit launch whose queue is no longer than the machine is wide?  Measures, on the C2 scene, incoherent rays (random origins in the atrium,
random directions: what waves 1+ look like) in batches of 64 .. 1 M through the ray-query seam (lumen_mi_query_closest -> lm_k_query_closest_raw, the per-lane traversal
of the wave kernels without the queue refill): per batch the traversal steps of the LONGEST ray and the mean (counting build), and — under
`rocprofv3 --kernel-trace --stats -- python3 tools/step_latency.py` — the kernel's duration.  One wavefront alone gives the UNLOADED latency of one dependent traversal step
(duration / longest chain); the larger batches show where the launch time stops being that chain and starts being throughput."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from lumenrenderer_amd import LumenRendererMI, scenes

d = scenes.sponza_standin()
r = LumenRendererMI(); r.Init(depth=2, render_resolution=(64, 64), blend_output=False)
r.LoadSceneDescription(d)
assert r.TraceFrame()
allv = np.concatenate([np.asarray(p["vertices"], np.float32).reshape(-1, 12)[:, :3] for p in d.primitives]) * 0.008
lo, hi = allv.min(0), allv.max(0)
rng = np.random.default_rng(5)
for n in (64, 1024, 16384, 131072, 524288, 1048576):
    o = (lo + (hi - lo) * (0.1 + 0.8 * rng.random((n, 3)))).astype(np.float32)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    r.SetInstrumented(True)
    r.QueryClosest(o, v.astype(np.float32))
    c = r.GetCounters(50)
    r.SetInstrumented(False)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ip, uvt = r.QueryClosest(o, v.astype(np.float32)); best = min(best, time.perf_counter() - t0)
    steps_mean = (c[22] / 4.0 + c[21]) / n            # 4-wide node steps (child boxes / 4) + triangle tests, per ray
    print(f"n {n:8d}  hits {(uvt[:, 2] > 0).mean():.3f}  steps per ray: mean {steps_mean:7.1f}  longest {c[40]:5d}   host round trip {best * 1e3:8.3f} ms (includes copies: use the rocprof kernel time)", flush=True)
r.close()
