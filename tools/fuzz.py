"""Schedule-fuzzing campaign (GPU box): many seeds x schedules x window shapes, each compared bit for bit with the serial schedule.
python tools/fuzz.py [seeds per configuration]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
import itertools, time
import numpy as np
from lumenrenderer_amd import LumenRendererMI, tiles
from lumenrenderer_amd.scenes import sponza_standin

SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
desc = sponza_standin()
base = np.array(desc.instances[0]["transform"], np.float32).reshape(4, 4)


def run(W, H, depth, tuning, window, tile, sync_each, frames=8):
    r = LumenRendererMI(); r.Init(depth=depth, render_resolution=(W, H), blend_output=True)
    r.LoadSceneDescription(desc); r.SetBlendMode(True)
    if window: r.SetWindow(*window)
    if tile: r.SetTile(*tile)
    for k, v in tuning.items(): r.SetTuning(k, v)
    inst = r.m_Scene.m_MeshInstances
    for k in range(frames):
        if k % 3 != 2:
            m = base.copy(); m[1, 3] += 0.002 * k; inst[0].SetTransform(m)
        if k == 4: inst[1].SetEmissiveness(2, (9.0, 8.0, 7.0), 30.0)
        if k == 5:                                 # topology edit: cleared and refilled with a second light quad
            sc = r.m_Scene; sc.Clear()
            for n, dd in enumerate(desc.instances + [desc.instances[1]]):
                mi = sc.AddMesh(r.m_Meshes[dd["mesh"]])
                t = np.array(dd["transform"], np.float32).reshape(4, 4).copy(); t[0, 3] += 1.5 * (n >= len(desc.instances))
                mi.SetTransform(t); mi.SetEmissiveness(dd["emission_mode"], dd["override_radiance"], dd["scale"])
            inst = sc.m_MeshInstances
        if k in (3, 6): r.SetDepth(depth + (k == 3))        # the swap chain's parity changes mid-run (lazy reuse: pending history passes run / are dropped)
        c = desc.camera
        r.SetCamera((c["position"][0] + 0.01 * k, c["position"][1], c["position"][2]), c["right"], c["up"], c["forward"], c["fov"])
        assert r.TraceFrameAsync()
        if sync_each: r.Synchronize()
    r.Synchronize()
    out = (r.GetRadiance().copy(), r.GetChannel(0).copy(), r.GetChannel(1).copy(), tuple(r.GetCounters()[:12]))
    r.close()
    return out


shapes = [(1280, 720, 5, None, None), (1280, 720, 6, None, None), (2560, 1440, 6, None, None)]
t8 = tiles.tile_rect(1, 8, 2560, 1440); shapes.append((2560, 1440, 5, tiles.window_rect(t8, 2560, 1440), t8))
schedules = [{}, {"pick_ahead": 0}, {"shadow_on_wave": 1}, {"tail_below": 0}, {"tail_below": 1 << 30}, {"tail_below": 60000, "tail_lanes": 64},
             {"tail_pair": 1, "tail_below": 1 << 30}, {"tail_pair": 1, "tail_below": 60000, "tail_lanes": 16},
             {"lazy_reuse": 1}, {"lazy_reuse": 1, "pick_ahead": 0, "tail_below": 0}, {"lazy_reuse": 1, "wave_streams": 2}, {"lazy_reuse": 0},
             {"tail_repack": 1}, {"tail_repack": 1, "lazy_reuse": 1, "tail_below": 1 << 30}, {"gpu_build": 1}, {"gpu_build": 1, "lazy_reuse": 1, "refit": 0}]        # round 4's options
bad = total = 0
t0 = time.time()
for (W, H, depth, window, tile) in shapes:
    ref = run(W, H, depth, {"single_stream": 1, "tail_below": 0, "pick_ahead": 0, "lazy_reuse": 0}, window, tile, True)       # serial, history passes with their frame
    for sched, seed in itertools.product(schedules, range(1, SEEDS + 1)):
        if W > 2000 and seed > max(2, SEEDS // 3): continue
        got = run(W, H, depth, dict(sched, fuzz=(0x9E3779B1 * (seed + 17 * len(sched)) + W) & 0x7fffffff), window, tile, False)
        ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3]
        total += 1
        if not ok:
            bad += 1
            print("MISMATCH", W, H, depth, window, sched, seed, [int(np.sum(a != b)) for a, b in zip(got[:3], ref[:3])])
print(f"{total} fuzzed runs, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
