"""Per-ray traversal-step histogram of one instrumented TraceFrame (GPU box): python tools/hist.py [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import sys, numpy as np
from lumenrenderer_amd import LumenRendererMI
from lumenrenderer_amd.scenes import sponza_standin
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r = LumenRendererMI(); r.Init(depth=depth, render_resolution=(2560, 1440), blend_output=True)
r.LoadSceneDescription(sponza_standin())
r.SetInstrumented(True); r.TraceFrame()
c = r.GetCounters(48)
print("depth", depth, "rays closest/nee/restir", c[0], c[1], c[2], "per wave", c[4:4 + depth])
print("hist log2(steps):", {f"<{2 ** (k + 1)}": int(c[24 + k]) for k in range(16) if c[24 + k]})
print("max steps per ray", c[40])
print("stack pushes: %d in LDS, %d spilled to global (%.3f %%)" % (c[45], c[46], 100.0 * c[46] / max(1, c[45] + c[46])))
print("lane occupancy: node steps %.3f (%d wave-issues), triangle tests %.3f (%d wave-issues)" % (c[41] / max(1, c[42]), c[42] // 64, c[43] / max(1, c[44]), c[44] // 64))
r.close()
