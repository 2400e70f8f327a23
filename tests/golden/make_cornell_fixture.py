#!/usr/bin/env python3
"""Container-only: reads the reference's own Cornell-box asset (data, not source) through lumenrenderer_amd.gltf and
stores the numbers (vertices incl. generated tangents, indices, material factors, instance transforms) as
tests/golden/cornell_box.npz, so the GPU box (which has no /root/reference) can build the same scene."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from lumenrenderer_amd.gltf import load_gltf
from lumenrenderer_amd.scenes import scene_to_npz
src = "/root/reference/Lumen_Engine/Sandbox/assets/models/CornellBox/scene.gltf"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_box.npz")
d = load_gltf(src)
scene_to_npz(d, dst)
print("triangles", d.triangle_count(), "primitives", len(d.primitives), "materials", len(d.materials), "->", dst, os.path.getsize(dst), "bytes")
