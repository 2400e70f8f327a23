"""Cost of a topology edit per frame (GPU box): the scene is cleared and refilled every frame with one more / one fewer instance;
instance-level assembly + GPU refit against the full host SAH rebuild.  python tools/topology.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import time
import numpy as np
from lumenrenderer_amd import LumenRendererMI
from lumenrenderer_amd.scenes import sponza_standin
for assemble in (1, 0):
    r = LumenRendererMI(); r.Init(depth=6, render_resolution=(2560, 1440), blend_output=False)
    desc = sponza_standin()
    r.LoadSceneDescription(desc); r.SetTuning("assemble", assemble)
    r.TraceFrame(); r.TraceFrame()

    def refill(k):
        sc = r.m_Scene
        sc.Clear()
        for i, inst in enumerate(desc.instances):
            mi = sc.AddMesh(r.m_Meshes[inst["mesh"]]); mi.SetTransform(inst["transform"])
            mi.SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])
        if k % 2:                                            # a second light quad on odd frames
            inst = desc.instances[1]
            t = np.array(inst["transform"], np.float32).reshape(4, 4).copy(); t[0, 3] += 2.0
            mi = sc.AddMesh(r.m_Meshes[inst["mesh"]]); mi.SetTransform(t); mi.SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])

    n = 10 if assemble else 4
    for k in range(2): refill(k); r.TraceFrameAsync()
    r.Synchronize(); t0 = time.perf_counter()
    for k in range(n):
        refill(k); r.TraceFrameAsync()
    r.Synchronize()
    print(f"assemble={assemble}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per TraceFrame with a topology edit before every frame, bvh {r.GetBvhInfo()}, assemblies {r.GetCounters(52)[51]}")
    r.close()
