#!/usr/bin/env python3
"""Container-only: the reference Sandbox's DEFAULT model (Sandbox/src/AppConfigDefaults.h:11: LowpolyRoom/scene.glb — data, not source) ingested
through lumenrenderer_amd.gltf and stored as numbers in tests/golden/ref_lowpoly_room.npz: vertices incl. generated tangents, indices, material factors
(three emissive materials), the one 512 x 512 base-colour map, instance transforms — and 64 camera poses produced by the reference's own Camera class
(lowpoly_camera.cpp, linked with the reference's Camera.cpp as it lies): pose 0 = Application.cpp:145-146, then OutputLayer.cpp's "W held + mouse drag"."""
import os, subprocess, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from lumenrenderer_amd.gltf import load_gltf
from lumenrenderer_amd.scenes import scene_to_npz
L = "/root/reference/Lumen_Engine/Lumen"
exe = "/tmp/lumen_lowpoly_camera"
subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-DNDEBUG", "-w", "-DGLM_ENABLE_EXPERIMENTAL", f"-I{L}/vendor/glm", f"-I{L}/src",
                       os.path.join(here, "lowpoly_camera.cpp"), f"{L}/src/Lumen/Renderer/Camera.cpp", "-o", exe])
poses = np.asarray([[float(x) for x in line.split()] for line in subprocess.check_output([exe], text=True).splitlines()], np.float32)
assert poses.shape == (64, 12)
d = load_gltf("/root/reference/Lumen_Engine/Sandbox/assets/models/LowpolyRoom/scene.glb")
p = poses[0]
d.set_camera(p[0:3], p[3:6], p[6:9], p[9:12], 90.0)
dst = os.path.join(here, "ref_lowpoly_room.npz")
scene_to_npz(d, dst, textures=True)
z = dict(np.load(dst)); z["camera_poses"] = poses
np.savez_compressed(dst, **z)
print("triangles", d.triangle_count(), "primitives", len(d.primitives), "materials", len(d.materials), "emissive materials",
      sum(1 for m in d.materials if any(m["emission"])), "textures", [t["pixels"].shape for t in d.textures], "->", dst, os.path.getsize(dst), "bytes")
print("pose 0:", poses[0])
