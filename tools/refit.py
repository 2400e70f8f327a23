"""Cost of a moving instance per frame (GPU box): GPU refit vs full host rebuild.  python tools/refit.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import time, numpy as np
from lumenrenderer_amd import LumenRendererMI
from lumenrenderer_amd.scenes import sponza_standin
for refit in (1, 0):
    r = LumenRendererMI(); r.Init(depth=6, render_resolution=(2560, 1440), blend_output=False)
    desc = sponza_standin(); base = np.array(desc.instances[0]["transform"], np.float32).reshape(4, 4)
    r.LoadSceneDescription(desc); r.SetTuning("refit", refit)
    inst = r.m_Scene.m_MeshInstances[0]
    r.TraceFrame(); r.TraceFrame()
    def run(move, n=8 if refit else 3):
        r.Synchronize(); t0 = time.perf_counter()
        for k in range(n):
            if move:
                m = base.copy(); m[1, 3] += 0.001 * (k + 1)
                inst.SetTransform(m)
            r.TraceFrameAsync()
        r.Synchronize(); return (time.perf_counter() - t0) / n * 1e3
    still, moving = run(False), run(True)
    print(f"refit={refit}: static {still:.2f} ms/TraceFrame, one instance moving every frame {moving:.2f} ms/TraceFrame, bvh {r.GetBvhInfo()}")
    r.close()
