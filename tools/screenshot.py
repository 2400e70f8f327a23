"""A picture of what the benchmark renders (GPU box): the C2 scene at 960x540, depth 6, 256 blended frames -> gpurun_out/<name>.png.
The benchmark's light (radiance x 50) saturates the 8-bit output almost everywhere; `scale` dims it for the picture only.
python tools/screenshot.py [name] [light scale, default 2.5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
from lumenrenderer_amd import LumenRendererMI
from lumenrenderer_amd.scenes import sponza_standin
name = sys.argv[1] if len(sys.argv) > 1 else "standin"
r = LumenRendererMI(); r.Init(depth=6, render_resolution=(960, 540), blend_output=True)
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
r.LoadSceneDescription(sponza_standin(light_scale=scale)); r.SetBlendMode(True)
for _ in range(256):
    r.TraceFrameAsync()
r.Synchronize()
os.makedirs("gpurun_out", exist_ok=True)
px = r.MakeScreenshot(os.path.join("gpurun_out", name + ".png"), gamma=1.0)          # the output is sRGB already; gamma 1 keeps it as rendered
print(px.shape, px[..., :3].mean())
r.close()
