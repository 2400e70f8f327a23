"""Shared test plumbing: replay one SceneDescription into the product (C ABI) and into the oracle."""
import os
import numpy as np

from oracle_lib import Oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TEX = ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
        "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture")
_ORC = {"diffuse_texture": "tex_diffuse", "normal_map": "tex_normal", "metallic_roughness_texture": "tex_metal_rough", "emissive_texture": "tex_emissive",
        "transmission_texture": "tex_transmission", "clearcoat_texture": "tex_clearcoat", "clearcoat_roughness_texture": "tex_clearcoat_rough", "tint_texture": "tex_tint",
        "transmission_factor": "transmission", "clearcoat_factor": "clearcoat", "clearcoat_roughness_factor": "clearcoat_roughness", "index_of_refraction": "ior",
        "specular_factor": "specular", "specular_tint_factor": "specular_tint", "subsurface_factor": "subsurface", "luminance": "luminance",
        "anisotropic": "anisotropic", "sheen_factor": "sheen", "sheen_tint_factor": "sheen_tint", "metallic_factor": "metallic", "roughness_factor": "roughness",
        "tint_factor": "tint", "transmittance": "transmittance", "diffuse_color": "diffuse_color", "emission": "emission"}


def oracle_from(desc, width, height, depth, blend=False, threads=None, window=None):
    o = Oracle(threads)
    tex = [o.add_texture(t["pixels"], t["srgb"]) for t in desc.textures]
    mats = []
    for m in desc.materials:
        kw = {_ORC[k]: (tex[v] if k in _TEX else v) for k, v in m.items()}
        mats.append(o.add_material(**kw))
    prims = [o.add_primitive(p["vertices"], p["indices"], mats[p["material"]]) for p in desc.primitives]
    meshes = [o.add_mesh([prims[i] for i in m]) for m in desc.meshes]
    for inst in desc.instances:
        o.add_instance(meshes[inst["mesh"]], inst["transform"], inst["emission_mode"], inst["override_radiance"], inst["scale"],
                       mats[inst["override_material"]] if inst["override_material"] >= 0 else -1)
    c = desc.camera
    o.set_camera(c["position"], c["right"], c["up"], c["forward"], c["fov"])
    o.set_resolution(width, height); o.set_depth(depth); o.set_blend(blend)
    if window:
        o.set_window(*window)
    return o


def product_from(desc, width, height, depth, blend=False, window=None, device=0, tuning=None):
    from lumenrenderer_amd import LumenRendererMI
    r = LumenRendererMI()
    r.Init(depth=depth, render_resolution=(width, height), blend_output=blend, device=device)
    r.LoadSceneDescription(desc)
    if blend:
        r.SetBlendMode(True)
    if window:
        r.SetWindow(*window)
    for k, v in (tuning or {}).items():
        r.SetTuning(k, v)
    return r


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.sqrt(np.sum(b * b))
    return float(np.sqrt(np.sum((a - b) ** 2)) / den) if den > 0 else float(np.sqrt(np.sum((a - b) ** 2)))


def cornell():
    from lumenrenderer_amd.scenes import cornell_box
    return cornell_box(fixture=os.path.join(GOLDEN, "cornell_box.npz"))


def random_soup(n_tris, seed, extent=10.0, size=1.0):
    """A SceneDescription holding a random triangle soup (one primitive + one emissive quad so that frames render)."""
    from lumenrenderer_amd.scenes import SceneDescription, interleave, generate_tangents_fast
    rng = np.random.default_rng(seed)
    d = SceneDescription()
    m = d.add_material(diffuse_color=(0.7, 0.7, 0.7, 1), metallic_factor=0.0, roughness_factor=0.8)
    c = rng.uniform(-extent, extent, (n_tris, 1, 3)); pos = (c + rng.uniform(-size, size, (n_tris, 3, 3))).reshape(-1, 3).astype(np.float32)
    e1, e2 = pos[1::3] - pos[0::3], pos[2::3] - pos[0::3]
    n = np.cross(e1, e2); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    nrm = np.repeat(n, 3, axis=0).astype(np.float32)
    uv = np.tile(np.float32([[0, 0], [1, 0], [0, 1]]), (n_tris, 1))
    idx = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    tang = generate_tangents_fast(pos, nrm, uv, idx)
    d.add_instance(d.add_mesh([d.add_primitive(interleave(pos, uv, nrm, tang), idx.ravel(), m)]))
    return d


def build_c_example(tmp_path):
    """examples/render_scene.c compiled as strict C99 against include/lumen_mi.h and linked with the product library."""
    import subprocess
    exe = str(tmp_path / "render_scene")
    libdir = os.path.join(ROOT, "lumenrenderer_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "render_scene.c"), "-o", exe, "-L" + libdir, "-llumen_mi", "-lm", "-Wl,-rpath," + libdir]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


def build_sandbox_driver(tmp_path):
    """examples/sandbox_driver.cpp (the Sandbox call sequence through include/lumen_mi_renderer.hpp) built against the minimal interface
    headers of examples/sandbox_min/ — no reference tree, no glm needed — and linked with the product library."""
    import subprocess
    exe = str(tmp_path / "sandbox_driver")
    libdir = os.path.join(ROOT, "lumenrenderer_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "examples", "sandbox_min"),
           os.path.join(ROOT, "examples", "sandbox_driver.cpp"), "-o", exe, "-L" + libdir, "-llumen_mi", "-lz", "-Wl,-rpath," + libdir]      # -lz: the minimal tree's .ollad reader inflates PNG images
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


def run_distributed(cmd, env, timeout=600, attempts=3):
    """subprocess.run of a `python -m torch.distributed.run ... --master-port P ...` command line; when the launcher cannot listen on P because another job on a shared host
    holds it (EADDRINUSE), the command runs again on a port the kernel has just handed out."""
    import socket, subprocess
    cmd = list(cmd)
    for _ in range(attempts):
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        if res.returncode == 0 or not ("EADDRINUSE" in res.stderr or "address already in use" in res.stderr.lower()) or "--master-port" not in cmd:
            break
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        cmd[cmd.index("--master-port") + 1] = str(port)
    return res
