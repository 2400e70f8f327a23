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
# round 5: three more of the reference's samples, WITH their texels — an emissive MATERIAL on a real mesh (EmissiveSphere: 1 472 triangle lights out of FindEmissives), the eight-material
# box.glb, and Glass/scene.gltf (77 124 triangles, alpha-blended materials that camera rays pass through and shadow rays do not, one emissive material = 15 359 triangle lights, two 512^2 maps)
for src, name, tex in (("cube/Cube.gltf", "ref_cube_textured.npz", True), ("CesiumMilkTruck/glTF/CesiumMilkTruck.gltf", "ref_milk_truck.npz", True),
                       ("EmissiveSphere/EmissiveSphere.gltf", "ref_emissive_sphere.npz", True), ("box/box.glb", "ref_box.npz", True), ("Glass/scene.gltf", "ref_glass.npz", True)):
    d = load_gltf(os.path.join(root, src))
    dst = os.path.join(here, name)
    scene_to_npz(d, dst, textures=tex)
    print(src, "triangles", d.triangle_count(), "primitives", len(d.primitives), "materials", len(d.materials), "textures",
          [t["pixels"].shape for t in d.textures], "->", name, os.path.getsize(dst), "bytes")
