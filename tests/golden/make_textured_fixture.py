#!/usr/bin/env python3
"""Container-only: ingests two small textured sample assets of the reference (data, not source) — the textured cube
(Sandbox/assets/models/cube, three PNG textures) and the Cesium milk truck (node hierarchy, several materials, a JPEG-free
PNG texture) — through lumenrenderer_amd.gltf and stores the decoded numbers (vertices incl. generated tangents, indices,
material factors, RGBA8 texels, instance transforms) as .npz, so the GPU box (no /root/reference) can build the same scenes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from lumenrenderer_amd.gltf import load_gltf
from lumenrenderer_amd.scenes import scene_to_npz
root = "/root/reference/Lumen_Engine/Sandbox/assets/models"
here = os.path.dirname(os.path.abspath(__file__))
for src, name in (("cube/Cube.gltf", "ref_cube_textured.npz"), ("CesiumMilkTruck/glTF/CesiumMilkTruck.gltf", "ref_milk_truck.npz")):
    d = load_gltf(os.path.join(root, src))
    dst = os.path.join(here, name)
    scene_to_npz(d, dst)
    print(src, "triangles", d.triangle_count(), "primitives", len(d.primitives), "materials", len(d.materials), "textures",
          [t["pixels"].shape for t in d.textures], "->", name, os.path.getsize(dst), "bytes")
