"""Scene descriptions (caller side of the plugin boundary) and the procedural benchmark scenes.

A ``SceneDescription`` is plain data — what ``SceneManager::LoadGLTF`` would feed through the renderer factories
(reference: Lumen/src/Lumen/ModelLoading/SceneManager.cpp:42-129).  It can be replayed into the product
(``LumenRendererMI.LoadSceneDescription``) or, in tests, into the CPU oracle.

Scenes:
  * ``cornell_box``      — the reference's own asset (Sandbox/assets/models/CornellBox/scene.gltf), from a glTF path or
                           from the committed fixture tests/golden/cornell_box.npz;
  * ``sponza_standin``   — ``Sponza.bin`` is missing from the reference mount, so C2/C3/C4 run on a procedural atrium with
                           the same statistics (262 267 triangles, 103 primitives, 25 materials, 16-bit indices, some
                           alpha-masked materials, object-space AABB of the real accessors, node scale 0.008);
  * ``foliage_stress``   — C5 deep-BVH stress (Moana is not part of the reference): instanced-then-flattened plants.
"""
import math

import numpy as np

TEXTURE_FIELDS = ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
                  "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture")


class SceneDescription:
    def __init__(self):
        self.textures, self.materials, self.primitives, self.meshes, self.instances = [], [], [], [], []
        self.camera = dict(position=(0, 0, 0), right=(-1, 0, 0), up=(0, 1, 0), forward=(0, 0, 1), fov=90.0)
        # default textures exactly as LumenPTModelConverter::SetRendererRef makes them (LumenPTModelConverter.cpp:320-333)
        self.tex_white = self.add_texture(np.array([[[255, 255, 255, 255]]], np.uint8), srgb=True)
        self.tex_metal_rough = self.add_texture(np.array([[[255, 255, 255, 255]]], np.uint8), srgb=False)
        self.tex_normal = self.add_texture(np.array([[[128, 128, 255, 0]]], np.uint8), srgb=False)
        self.tex_emissive = self.add_texture(np.array([[[255, 255, 255, 255]]], np.uint8), srgb=True)

    def add_texture(self, pixels, srgb):
        self.textures.append(dict(pixels=np.ascontiguousarray(pixels, np.uint8), srgb=bool(srgb)))
        return len(self.textures) - 1

    def add_material(self, **kw):
        """Field names of lumen_mi_material_data; defaults follow the .ollad ingest path the reference takes
        (LumenPTModelConverter.cpp:336-560,141-200): tint 0, transmittance 0, luminance 1, ior 1, roughness >= 0.01."""
        m = dict(diffuse_color=(1, 1, 1, 1), emission=(0, 0, 0), diffuse_texture=self.tex_white, normal_map=self.tex_normal,
                 metallic_roughness_texture=self.tex_metal_rough, emissive_texture=self.tex_emissive, transmission_texture=self.tex_white,
                 clearcoat_texture=self.tex_white, clearcoat_roughness_texture=self.tex_white, tint_texture=self.tex_white,
                 transmission_factor=0.0, clearcoat_factor=0.0, clearcoat_roughness_factor=0.0, index_of_refraction=1.0, specular_factor=0.0,
                 specular_tint_factor=0.0, subsurface_factor=0.0, luminance=1.0, anisotropic=0.0, sheen_factor=0.0, sheen_tint_factor=0.0,
                 metallic_factor=1.0, roughness_factor=1.0, tint_factor=(0, 0, 0), transmittance=(0, 0, 0))
        m.update(kw)
        m["roughness_factor"] = max(0.01, float(m["roughness_factor"]))
        self.materials.append(m)
        return len(self.materials) - 1

    def add_primitive(self, vertices, indices, material, index_size=None):
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 12)
        i = np.ascontiguousarray(indices, np.uint32).ravel()
        if index_size is None:
            index_size = 2 if v.shape[0] < 65536 else 4
        self.primitives.append(dict(vertices=v, indices=i, material=material, index_size=index_size))
        return len(self.primitives) - 1

    def add_mesh(self, prims):
        self.meshes.append(list(prims))
        return len(self.meshes) - 1

    def add_instance(self, mesh, transform=None, emission_mode=0, override_radiance=(0, 0, 0), scale=1.0, override_material=-1):
        t = np.eye(4, dtype=np.float32) if transform is None else np.asarray(transform, np.float32).reshape(4, 4)
        self.instances.append(dict(mesh=mesh, transform=t, emission_mode=emission_mode, override_radiance=tuple(override_radiance),
                                   scale=float(scale), override_material=override_material))
        return len(self.instances) - 1

    def set_camera(self, position, right, up, forward, fov=90.0):
        self.camera = dict(position=tuple(position), right=tuple(right), up=tuple(up), forward=tuple(forward), fov=float(fov))

    def triangle_count(self):
        return sum(len(self.primitives[p]["indices"]) // 3 for inst in self.instances for p in self.meshes[inst["mesh"]])


# ---------------------------------------------------------------------------------------------------------------------
# vertex helpers
# ---------------------------------------------------------------------------------------------------------------------
_DEFAULT_UV = np.array([[1.0, 1.0], [0.0, 1.0], [1.0, 0.0]], np.float32)


def generate_tangents(pos, normal, uv, indices):
    """LumenPTModelConverter::GenerateTangentBinary (LumenPTModelConverter.cpp:734-900): per-triangle tangent from the
    UV parametrisation (fixed default UVs when absent/degenerate), Gram-Schmidt against the vertex normal, w = +1; a
    vertex keeps the tangent of the last triangle that touches it."""
    pos = np.asarray(pos, np.float32); normal = np.asarray(normal, np.float32)
    tang = np.zeros((pos.shape[0], 4), np.float32)
    eps = np.finfo(np.float32).eps
    idx = np.asarray(indices).reshape(-1, 3)
    for tri in idx:
        v0, v1, v2 = pos[tri[0]], pos[tri[1]], pos[tri[2]]
        if uv is None:
            uv0, uv1, uv2 = _DEFAULT_UV
        else:
            uv0, uv1, uv2 = uv[tri[0]], uv[tri[1]], uv[tri[2]]
            len2 = lambda d: np.sqrt(np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])), dtype=np.float32)      # glm::length of a vec2
            d0, d1, d2 = len2(uv0 - uv1), len2(uv0 - uv2), len2(uv2 - uv1)
            if d0 < eps or d1 < eps or d2 < eps:
                uv0, uv1, uv2 = _DEFAULT_UV
        dp1, dp2 = v1 - v0, v2 - v0
        duv1, duv2 = uv1 - uv0, uv2 - uv0
        cross = np.float32(duv1[0] * duv2[1] - duv1[1] * duv2[0])
        if cross == 0:
            uv0, uv1, uv2 = _DEFAULT_UV
            duv1, duv2 = uv1 - uv0, uv2 - uv0
        t = (duv2[1] * dp1 - duv1[1] * dp2) / np.float32(duv1[0] * duv2[1] - duv2[0] * duv1[1])
        for k in range(3):
            ng = _glm_normalize3(normal[tri[k]])
            tt = _glm_normalize3(t - ng * _glm_dot3(ng, t))
            tang[tri[k]] = (tt[0], tt[1], tt[2], 1.0)
    return tang


def _glm_dot3(a, b):
    """glm::dot for vec3 in float: the three products, summed left to right (glm/detail/func_geometric.inl compute_dot<vec<3,...>>)"""
    t = (np.asarray(a, np.float32) * np.asarray(b, np.float32)).astype(np.float32)
    return np.float32(np.float32(t[0] + t[1]) + t[2])


def _glm_normalize3(v):
    """glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x) in float — a multiplication by the rounded reciprocal, not a division by the length
    (the two differ in the last bit; pinned against the reference converter's own .ollad output, tests/test_cpu_host.py)"""
    v = np.asarray(v, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1.0) / np.sqrt(_glm_dot3(v, v), dtype=np.float32)
    return (v * inv).astype(np.float32)


def generate_tangents_fast(pos, normal, uv, indices):
    """Vectorised equivalent for the large procedural meshes (valid UVs everywhere; 'last triangle wins' kept)."""
    pos = np.asarray(pos, np.float32); normal = np.asarray(normal, np.float32); uv = np.asarray(uv, np.float32)
    idx = np.asarray(indices).reshape(-1, 3)
    v0, v1, v2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    uv0, uv1, uv2 = uv[idx[:, 0]], uv[idx[:, 1]], uv[idx[:, 2]]
    dp1, dp2, duv1, duv2 = v1 - v0, v2 - v0, uv1 - uv0, uv2 - uv0
    det = duv1[:, 0] * duv2[:, 1] - duv2[:, 0] * duv1[:, 1]
    bad = det == 0
    if bad.any():
        d1, d2 = _DEFAULT_UV[1] - _DEFAULT_UV[0], _DEFAULT_UV[2] - _DEFAULT_UV[0]
        duv1[bad], duv2[bad] = d1, d2
        det = duv1[:, 0] * duv2[:, 1] - duv2[:, 0] * duv1[:, 1]
    t = (duv2[:, 1:2] * dp1 - duv1[:, 1:2] * dp2) / det[:, None]
    tang = np.zeros((pos.shape[0], 4), np.float32); tang[:, 0] = 1.0; tang[:, 3] = 1.0
    for k in range(3):
        ng = normal[idx[:, k]]
        ng = ng / np.linalg.norm(ng, axis=1, keepdims=True)
        tt = t - ng * np.sum(ng * t, axis=1, keepdims=True)
        nrm = np.linalg.norm(tt, axis=1, keepdims=True)
        ok = nrm[:, 0] > 0
        tt = np.where(ok[:, None], tt / np.where(nrm == 0, 1, nrm), np.array([1.0, 0.0, 0.0], np.float32))
        tang[idx[:, k], :3] = tt          # numpy keeps the last write for duplicate indices = "last triangle wins"
    return tang.astype(np.float32)


def interleave(pos, uv, normal, tangent):
    n = len(pos)
    v = np.zeros((n, 12), np.float32)
    v[:, 0:3] = pos
    if uv is not None:
        v[:, 3:5] = uv
    if normal is not None:
        v[:, 5:8] = normal
    if tangent is not None:
        v[:, 8:12] = tangent
    return v


def camera_from_rotation(position, rotation_matrix, fov=90.0):
    """right/up/forward = columns 0/1/2 of the rotation matrix (Camera.cpp:128-140)."""
    r = np.asarray(rotation_matrix, np.float32).reshape(3, 3)
    return dict(position=tuple(np.asarray(position, np.float32)), right=tuple(r[:, 0]), up=tuple(r[:, 1]), forward=tuple(r[:, 2]), fov=float(fov))


# ---------------------------------------------------------------------------------------------------------------------
# Cornell box (C1)
# ---------------------------------------------------------------------------------------------------------------------
def cornell_box(path=None, fixture=None):
    """C1.  ``path`` = the reference's scene.gltf (build container), else ``fixture`` = tests/golden/cornell_box.npz.
    Camera per SURVEY.md §8 d2: eye (0,1,3.4), rotation 180 deg about +Y => right (-1,0,0), up (0,1,0), forward (0,0,-1)."""
    if path is not None:
        from .gltf import load_gltf
        desc = load_gltf(path)
    else:
        desc = scene_from_npz(fixture)
    desc.set_camera((0.0, 1.0, 3.4), (-1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, -1.0), 90.0)
    return desc


def lowpoly_room(fixture):
    """The reference Sandbox's DEFAULT model (Sandbox/src/AppConfigDefaults.h:11, LowpolyRoom/scene.glb: 20 501 triangles, 10 primitives, 10 materials of which
    three are emissive — the only lights: no OVERRIDE instance —, one 512 x 512 sRGB base-colour map) from the committed numbers of
    tests/golden/ref_lowpoly_room.npz (tests/golden/make_lowpoly_fixture.py), with the camera Application.cpp:145-146 sets.  ``desc.camera_poses`` holds 64 poses of
    the reference's own Camera class driven by OutputLayer.cpp's input handling (W held, mouse drag): rows of eye, right, up, forward."""
    desc = scene_from_npz(fixture)
    desc.camera_poses = np.load(fixture)["camera_poses"]
    return desc


def lowpoly_camera_pose(desc, k):
    """Pose k (0..63) of the LowpolyRoom walk as SetCamera arguments."""
    p = desc.camera_poses[k]
    return (p[0:3].copy(), p[3:6].copy(), p[6:9].copy(), p[9:12].copy(), 90.0)


_MAT_SCALARS = ("transmission_factor", "clearcoat_factor", "clearcoat_roughness_factor", "index_of_refraction", "specular_factor", "specular_tint_factor",
                "subsurface_factor", "luminance", "anisotropic", "sheen_factor", "sheen_tint_factor", "metallic_factor", "roughness_factor")


def scene_to_npz(desc, path, textures=False):
    """Numbers only: vertices, indices, material factors, instance transforms; with ``textures`` also the RGBA8 texels of every texture
    behind the four defaults, their sRGB flags and each material's eight texture slots (without it every slot reads its default texture)."""
    out = {"n_prims": np.int64(len(desc.primitives)), "n_mats": np.int64(len(desc.materials))}
    if textures:
        out["n_tex"] = np.int64(len(desc.textures))
        for i, t in enumerate(desc.textures[4:], 4):
            out[f"t{i}_px"], out[f"t{i}_srgb"] = t["pixels"], np.int64(t["srgb"])
        for i, m in enumerate(desc.materials):
            out[f"m{i}_tex"] = np.asarray([m[k] for k in TEXTURE_FIELDS], np.int64)
    c = desc.camera
    out["camera"] = np.asarray(list(c["position"]) + list(c["right"]) + list(c["up"]) + list(c["forward"]) + [c["fov"]], np.float32)
    for i, p in enumerate(desc.primitives):
        out[f"p{i}_v"], out[f"p{i}_i"], out[f"p{i}_m"], out[f"p{i}_s"] = p["vertices"], p["indices"], np.int64(p["material"]), np.int64(p["index_size"])
    for i, m in enumerate(desc.materials):
        out[f"m{i}_color"], out[f"m{i}_emission"] = np.asarray(m["diffuse_color"], np.float32), np.asarray(m["emission"], np.float32)
        out[f"m{i}_scalars"] = np.asarray([m[k] for k in _MAT_SCALARS], np.float32)
        out[f"m{i}_tint"], out[f"m{i}_transmittance"] = np.asarray(m["tint_factor"], np.float32), np.asarray(m["transmittance"], np.float32)
    out["meshes"] = np.asarray([len(m) for m in desc.meshes], np.int64)
    out["mesh_prims"] = np.asarray([p for m in desc.meshes for p in m], np.int64)
    out["inst_mesh"] = np.asarray([i["mesh"] for i in desc.instances], np.int64)
    out["inst_xf"] = np.asarray([i["transform"] for i in desc.instances], np.float32)
    out["inst_mode"] = np.asarray([i["emission_mode"] for i in desc.instances], np.int64)
    out["inst_rad"] = np.asarray([i["override_radiance"] for i in desc.instances], np.float32)
    out["inst_scale"] = np.asarray([i["scale"] for i in desc.instances], np.float32)
    np.savez_compressed(path, **out)


def scene_from_npz(path):
    z = np.load(path)
    d = SceneDescription()
    for i in range(4, int(z["n_tex"]) if "n_tex" in z else 0):
        d.add_texture(z[f"t{i}_px"], bool(z[f"t{i}_srgb"]))
    for i in range(int(z["n_mats"])):
        kw = dict(zip(_MAT_SCALARS, [float(x) for x in z[f"m{i}_scalars"]]))
        if f"m{i}_tex" in z:
            kw.update(zip(TEXTURE_FIELDS, [int(x) for x in z[f"m{i}_tex"]]))
        d.add_material(diffuse_color=tuple(z[f"m{i}_color"]), emission=tuple(z[f"m{i}_emission"]), tint_factor=tuple(z[f"m{i}_tint"]),
                       transmittance=tuple(z[f"m{i}_transmittance"]), **kw)
    for i in range(int(z["n_prims"])):
        d.add_primitive(z[f"p{i}_v"], z[f"p{i}_i"], int(z[f"p{i}_m"]), int(z[f"p{i}_s"]))
    k = 0
    for n in z["meshes"]:
        d.add_mesh([int(p) for p in z["mesh_prims"][k:k + int(n)]]); k += int(n)
    for j in range(len(z["inst_mesh"])):
        d.add_instance(int(z["inst_mesh"][j]), z["inst_xf"][j], int(z["inst_mode"][j]), tuple(z["inst_rad"][j]), float(z["inst_scale"][j]))
    if "camera" in z:
        c = z["camera"]
        d.set_camera(tuple(c[0:3]), tuple(c[3:6]), tuple(c[6:9]), tuple(c[9:12]), float(c[12]))
    return d


# ---------------------------------------------------------------------------------------------------------------------
# procedural geometry
# ---------------------------------------------------------------------------------------------------------------------
class _Xorshift:
    """xorshift32 of the reference (RandomUtilities.cuh:10-13); seeds are fixed per scene."""

    def __init__(self, seed):
        self.s = np.uint32(seed)

    def u32(self):
        s = int(self.s)
        s ^= (s << 13) & 0xFFFFFFFF; s ^= s >> 17; s ^= (s << 5) & 0xFFFFFFFF
        self.s = np.uint32(s)
        return s

    def f(self):
        return self.u32() * 2.3283064365387e-10

    def uniform(self, a, b):
        return a + (b - a) * self.f()


def _grid(nu, nv, fn):
    """Parametric patch: fn(u, v) -> (pos[...,3], normal[...,3]); returns interleaved-ready arrays and indices."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float64), np.linspace(0, 1, nv + 1, dtype=np.float64), indexing="xy")
    pos, nrm = fn(u, v)
    pos = pos.reshape(-1, 3).astype(np.float32); nrm = nrm.reshape(-1, 3).astype(np.float32)
    uv = np.stack([u.ravel(), v.ravel()], 1).astype(np.float32)
    i0 = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).ravel()
    idx = np.stack([i0, i0 + 1, i0 + nu + 2, i0, i0 + nu + 2, i0 + nu + 1], 1).astype(np.uint32)
    return pos, uv, nrm, idx.reshape(-1, 3)


def _merge(parts):
    pos, uv, nrm, idx, base = [], [], [], [], 0
    for p, t, n, i in parts:
        pos.append(p); uv.append(t); nrm.append(n); idx.append(i + base); base += len(p)
    return np.concatenate(pos), np.concatenate(uv), np.concatenate(nrm), np.concatenate(idx)


def _quad(p0, eu, ev, nu=1, nv=1, flip=False):
    p0, eu, ev = np.asarray(p0, np.float64), np.asarray(eu, np.float64), np.asarray(ev, np.float64)
    n = np.cross(eu, ev); n = n / np.linalg.norm(n)

    def fn(u, v):
        pos = p0 + u[..., None] * eu + v[..., None] * ev
        return pos, np.broadcast_to(n, pos.shape)
    p, t, nn, i = _grid(nu, nv, fn)
    if flip:
        i = i[:, ::-1].copy(); nn = -nn
    return p, t, nn, i


def _box(lo, hi, n=1):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    X, Y, Z = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    faces = [(lo, Z, Y), (lo + X, Y, Z), (lo, X, Z), (lo + Y, Z, X), (lo, Y, X), (lo + Z, X, Y)]
    return _merge([_quad(p, a, b, n, n) for p, a, b in faces])


def _cylinder(center, radius, height, sides, rings, wobble=0.0):
    c = np.asarray(center, np.float64)

    def fn(u, v):
        a = u * 2 * math.pi
        r = radius * (1.0 + wobble * np.sin(v * math.pi * 6))
        pos = np.stack([c[0] + r * np.cos(a), c[1] + v * height, c[2] + r * np.sin(a)], -1)
        nrm = np.stack([np.cos(a), np.zeros_like(a), np.sin(a)], -1)
        return pos, nrm
    p, t, n, i = _grid(sides, rings, fn)
    return p, t, n, i[:, ::-1].copy()


def _arch(center, radius, thickness, depth, segs, axis=0):
    c = np.asarray(center, np.float64)
    parts = []
    for r, flip in ((radius, True), (radius + thickness, False)):
        def fn(u, v, r=r):
            a = u * math.pi
            if axis == 0:
                pos = np.stack([c[0] + r * np.cos(a), c[1] + r * np.sin(a), c[2] + (v - 0.5) * depth], -1)
                nrm = np.stack([np.cos(a), np.sin(a), np.zeros_like(a)], -1)
            else:
                pos = np.stack([c[0] + (v - 0.5) * depth, c[1] + r * np.sin(a), c[2] + r * np.cos(a)], -1)
                nrm = np.stack([np.zeros_like(a), np.sin(a), np.cos(a)], -1)
            return pos, nrm
        p, t, n, i = _grid(segs, 2, fn)
        if flip:
            i = i[:, ::-1].copy(); n = -n
        parts.append((p, t, n, i))
    return _merge(parts)


def _cloth(p0, width_vec, drop, nu, nv, amp, waves, phase):
    p0, wv = np.asarray(p0, np.float64), np.asarray(width_vec, np.float64)
    side = np.cross(wv / np.linalg.norm(wv), np.array([0.0, 1.0, 0.0]))

    def fn(u, v):
        off = amp * np.sin(u * waves * 2 * math.pi + phase) * (0.3 + 0.7 * v)
        pos = p0 + u[..., None] * wv + v[..., None] * np.array([0.0, -drop, 0.0]) + off[..., None] * side
        dd = amp * np.cos(u * waves * 2 * math.pi + phase) * waves * 2 * math.pi * (0.3 + 0.7 * v) / np.linalg.norm(wv)
        nrm = side[None, None, :] - dd[..., None] * (wv / np.linalg.norm(wv))
        nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
        return pos, nrm
    return _grid(nu, nv, fn)


def _blob(center, radius, nu, nv, rng, bumps=6):
    c = np.asarray(center, np.float64)
    ph = [(rng.uniform(1, 5), rng.uniform(1, 5), rng.uniform(0, 6.28), rng.uniform(0.02, 0.08)) for _ in range(bumps)]

    def fn(u, v):
        a, b = u * 2 * math.pi, (v - 0.5) * math.pi * 0.98
        r = radius * np.ones_like(a)
        for fa, fb, p, am in ph:
            r = r * (1.0 + am * np.sin(fa * a + p) * np.cos(fb * b))
        d = np.stack([np.cos(b) * np.cos(a), np.sin(b), np.cos(b) * np.sin(a)], -1)
        return c + r[..., None] * d, d
    p, t, n, i = _grid(nu, nv, fn)
    return p, t, n, i[:, ::-1].copy()


def _alpha_texture(rng, size=64):
    """Leaf-like alpha mask (alpha < 0.51 is cut, GPUExtractSurfaceData.cu:139)."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / (size - 1)
    img = np.zeros((size, size, 4), np.uint8)
    mask = np.zeros((size, size), bool)
    for _ in range(7):
        cx, cy, r = rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.12, 0.3)
        mask |= ((x - cx) ** 2 / (r * r) + (y - cy) ** 2 / (0.35 * r * r)) < 1.0
    img[..., 0] = (60 + 80 * y).astype(np.uint8); img[..., 1] = (120 + 100 * x).astype(np.uint8); img[..., 2] = 40
    img[..., 3] = np.where(mask, 255, 0)
    return img


def _add_part(desc, part, material, prims):
    pos, uv, nrm, idx = part
    tang = generate_tangents_fast(pos, nrm, uv, idx)
    prims.append(desc.add_primitive(interleave(pos, uv, nrm, tang), idx.ravel(), material))
    return len(idx)


SPONZA_TRIANGLES, SPONZA_PRIMITIVES, SPONZA_MATERIALS = 262267, 103, 25
SPONZA_AABB_MIN = np.array([-1920.9, -126.4, -1182.8])
SPONZA_AABB_MAX = np.array([1799.9, 1429.4, 1105.4])
SPONZA_NODE_SCALE = 0.008


def procedural_maps(seed, size=1024, tiles=6):
    """Seeded base-colour / normal / metal-roughness maps (RGBA8, size x size) for the textured variant of the stand-in: multi-octave value
    noise over a tile pattern with grout lines; the normal map is the gradient of the same height field; roughness follows a second noise
    field in the G channel (>= 1/255, as the reference's loader clamps it, LumenPTModelConverter.cpp:121-128), metallic in B.  Deterministic
    (PCG64 of ``seed``); tests/test_cpu_host.py pins a checksum."""
    g = np.random.Generator(np.random.PCG64(int(seed)))
    x = (np.arange(size, dtype=np.float64) + 0.5) / size

    def octave(n):                                                # n x n lattice, bilinear, wrapping
        lat = g.random((n, n))
        f = x * n - 0.5
        i0 = np.floor(f).astype(np.int64); t = f - i0
        i0 %= n; i1 = (i0 + 1) % n
        rows = lat[i0][:, i0] * (1 - t)[None, :] + lat[i0][:, i1] * t[None, :]
        rows1 = lat[i1][:, i0] * (1 - t)[None, :] + lat[i1][:, i1] * t[None, :]
        return rows * (1 - t)[:, None] + rows1 * t[:, None]

    def field():
        h = sum(octave(8 << k) * 0.5 ** k for k in range(5))
        return h / sum(0.5 ** k for k in range(5))
    h, h2 = field(), field()
    fu = (x * tiles) % 1.0
    grout = (fu < 0.04)[None, :] | (fu < 0.04)[:, None]
    height = np.where(grout, 0.15 * h, 0.35 + 0.65 * h)
    tint = 0.85 + 0.15 * g.random(3)
    albedo = np.where(grout, 0.45, 0.6 + 0.4 * h)[..., None] * tint[None, None, :]
    diffuse = np.empty((size, size, 4), np.uint8)
    diffuse[..., :3] = np.clip(np.rint(albedo * 255.0), 0, 255).astype(np.uint8); diffuse[..., 3] = 255
    du = (np.roll(height, -1, axis=1) - np.roll(height, 1, axis=1)) * 0.5 * size
    dv = (np.roll(height, -1, axis=0) - np.roll(height, 1, axis=0)) * 0.5 * size
    n = np.stack([-du * 0.012, -dv * 0.012, np.ones_like(du)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    normal = np.empty((size, size, 4), np.uint8)
    normal[..., :3] = np.clip(np.rint((n * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8); normal[..., 3] = 0
    mr = np.full((size, size, 4), 255, np.uint8)
    mr[..., 1] = np.clip(np.rint((0.45 + 0.55 * h2) * 255.0), 1, 255).astype(np.uint8)
    mr[..., 2] = np.where(grout, 160, 255).astype(np.uint8)
    return diffuse, normal, mr


def sponza_standin(extra_lights=0, light_radiance=(17.0, 12.0, 4.0), light_scale=50.0, textured=False, tex_size=1024):
    """C2 (and C3 with ``extra_lights=512`` quads = 1024 emissive triangles).  Deterministic: xorshift32 seeded 'SPON'.
    ``textured``: the 22 opaque / metal materials get seeded ``tex_size``^2 base-colour (sRGB), normal and metal-roughness maps
    (procedural_maps), so that surface extraction really performs its bilinear texture fetches (GPUExtractSurfaceData.cu:59-60,169-181);
    geometry, UVs, lights and camera are those of the untextured scene."""
    rng = _Xorshift(0x53504F4E)
    d = SceneDescription()
    lo, hi = SPONZA_AABB_MIN, SPONZA_AABB_MAX
    ext = hi - lo
    # ---- 25 materials: 20 opaque dielectrics / 2 metals / 2 alpha-masked / 1 light
    mats = []
    palette = [(0.72, 0.68, 0.6), (0.62, 0.58, 0.5), (0.55, 0.22, 0.18), (0.18, 0.3, 0.55), (0.2, 0.45, 0.22), (0.7, 0.6, 0.3), (0.4, 0.38, 0.36),
               (0.8, 0.78, 0.74), (0.5, 0.3, 0.2), (0.3, 0.3, 0.32)]
    for k in range(20):
        c = palette[k % len(palette)]
        j = 0.9 + 0.2 * rng.f()
        mats.append(d.add_material(diffuse_color=(min(1, c[0] * j), min(1, c[1] * j), min(1, c[2] * j), 1.0), metallic_factor=0.0,
                                   roughness_factor=0.35 + 0.6 * rng.f(), specular_factor=0.5 * rng.f()))
    mats.append(d.add_material(diffuse_color=(0.9, 0.7, 0.35, 1.0), metallic_factor=1.0, roughness_factor=0.25))
    mats.append(d.add_material(diffuse_color=(0.75, 0.76, 0.78, 1.0), metallic_factor=0.9, roughness_factor=0.4))
    for _ in range(2):
        tex = d.add_texture(_alpha_texture(rng), srgb=True)
        mats.append(d.add_material(diffuse_color=(1, 1, 1, 1), diffuse_texture=tex, metallic_factor=0.0, roughness_factor=0.8))
    light_mat = d.add_material(diffuse_color=(1, 1, 1, 1), emission=(1, 1, 1), metallic_factor=0.0, roughness_factor=1.0)
    mats.append(light_mat)
    assert len(mats) == SPONZA_MATERIALS
    if textured:
        for k in range(22):
            dm, nm, mr = procedural_maps(0x54455800 + k, tex_size, tiles=4 + k % 5)          # 'TEX' + material number
            m = d.materials[mats[k]]
            m["diffuse_texture"] = d.add_texture(dm, srgb=True)           # the .ollad path decodes base colour as sRGB (LumenPTModelConverter.cpp:130-133)
            m["normal_map"] = d.add_texture(nm, srgb=False)
            m["metallic_roughness_texture"] = d.add_texture(mr, srgb=False)
    prims, tri = [], 0
    opaque = mats[:20]
    fy = lo[1]                                                    # floor height
    # 1-6: floor, ceiling ring, four outer walls (tessellated so the BVH has real work)
    tri += _add_part(d, _quad((lo[0], fy, lo[2]), (0, 0, ext[2]), (ext[0], 0, 0), 64, 96), opaque[0], prims)
    tri += _add_part(d, _quad((lo[0], fy, lo[2]), (ext[0], 0, 0), (0, ext[1], 0), 96, 40), opaque[1], prims)
    tri += _add_part(d, _quad((lo[0], fy, hi[2]), (0, ext[1], 0), (ext[0], 0, 0), 40, 96), opaque[1], prims)
    tri += _add_part(d, _quad((lo[0], fy, lo[2]), (0, ext[1], 0), (0, 0, ext[2]), 40, 64), opaque[2], prims)
    tri += _add_part(d, _quad((hi[0], fy, lo[2]), (0, 0, ext[2]), (0, ext[1], 0), 64, 40), opaque[2], prims)
    # roof with a central opening: four strips
    ry = hi[1]
    ox0, ox1, oz0, oz1 = lo[0] + 0.3 * ext[0], hi[0] - 0.3 * ext[0], lo[2] + 0.35 * ext[2], hi[2] - 0.35 * ext[2]
    roof = _merge([_quad((lo[0], ry, lo[2]), (ext[0], 0, 0), (0, 0, oz0 - lo[2]), 48, 12), _quad((lo[0], ry, oz1), (ext[0], 0, 0), (0, 0, hi[2] - oz1), 48, 12),
                   _quad((lo[0], ry, oz0), (ox0 - lo[0], 0, 0), (0, 0, oz1 - oz0), 12, 12), _quad((ox1, ry, oz0), (hi[0] - ox1, 0, 0), (0, 0, oz1 - oz0), 12, 12)])
    tri += _add_part(d, roof, opaque[3], prims)
    # 7-8: upper gallery slabs on both long sides
    gy = fy + 0.42 * ext[1]
    for s, z0, z1 in ((0, lo[2], lo[2] + 0.27 * ext[2]), (1, hi[2] - 0.27 * ext[2], hi[2])):
        tri += _add_part(d, _box((lo[0], gy - 25, z0), (hi[0], gy, z1), 24), opaque[4 + s], prims)
    # 9-56: two storeys x two rows x 12 columns
    col_z = (lo[2] + 0.27 * ext[2], hi[2] - 0.27 * ext[2])
    col_x = np.linspace(lo[0] + 0.08 * ext[0], hi[0] - 0.08 * ext[0], 12)
    for storey, (y0, hgt, rad) in enumerate(((fy, gy - 25 - fy, 42.0), (gy, 0.42 * ext[1], 34.0))):
        for z in col_z:
            for x in col_x:
                shaft = _cylinder((x, y0 + 30, z), rad, hgt - 60, 40, 24, wobble=0.015)
                base = _box((x - rad * 1.3, y0, z - rad * 1.3), (x + rad * 1.3, y0 + 30, z + rad * 1.3), 2)
                cap = _box((x - rad * 1.4, y0 + hgt - 30, z - rad * 1.4), (x + rad * 1.4, y0 + hgt, z + rad * 1.4), 2)
                tri += _add_part(d, _merge([shaft, base, cap]), opaque[6 + storey], prims)
    # 57-78: arches between ground-floor columns (11 per row)
    for z in col_z:
        for a, b in zip(col_x[:-1], col_x[1:]):
            tri += _add_part(d, _arch(((a + b) / 2, gy - 25 - (b - a) / 2 - 20, z), (b - a) / 2 - 30, 25, 60, 40, axis=0), opaque[8], prims)
    # 79-86: eight hanging cloths (banners) across the nave
    for k in range(8):
        x = lo[0] + (0.12 + 0.1 * k) * ext[0]
        z0 = col_z[0] + 40
        tri += _add_part(d, _cloth((x, gy + 260, z0), (0, 0, (col_z[1] - col_z[0]) - 80), 420 + 40 * (k % 3), 64, 40, 28.0, 3 + k % 3, rng.uniform(0, 6.28)),
                         opaque[9 + (k % 4)], prims)
    # 87-94: eight vases (metal / stone) with 95-100: six masked plants on top of six of them
    vase_xy = [(lo[0] + (0.15 + 0.1 * k) * ext[0], (col_z[0] + 160) if k % 2 == 0 else (col_z[1] - 160)) for k in range(8)]
    for k, (x, z) in enumerate(vase_xy):
        tri += _add_part(d, _blob((x, fy + 85, z), 55.0, 48, 32, rng), mats[20 + (k % 2)] if k < 4 else opaque[13 + k % 3], prims)
    for k, (x, z) in enumerate(vase_xy[:6]):
        leaves = []
        for _ in range(160):
            a, tilt, r, h = rng.uniform(0, 6.28), rng.uniform(0.2, 1.2), rng.uniform(40, 120), rng.uniform(120, 260)
            eu = np.array([math.cos(a) * r, math.sin(tilt) * r * 0.6, math.sin(a) * r])
            ev = np.cross(eu, (0, 1, 0)); ev = ev / np.linalg.norm(ev) * r * 0.45
            leaves.append(_quad((x, fy + h, z), eu, ev, 1, 1))
        tri += _add_part(d, _merge(leaves), mats[22 + (k % 2)], prims)
    # 101: a relief ("lion head") on the far wall
    tri += _add_part(d, _blob((hi[0] - 230, fy + 0.35 * ext[1], (lo[2] + hi[2]) / 2), 140.0, 100, 60, rng, bumps=14), mats[21], prims)
    # 102: a basin in the middle of the nave
    tri += _add_part(d, _blob(((lo[0] + hi[0]) / 2, fy + 135, (lo[2] + hi[2]) / 2), 110.0, 96, 48, rng, bumps=4), opaque[17], prims)
    # 103: filler detail mesh (floor tiles border) sized so that the total is exactly the Sponza triangle count
    assert len(prims) == SPONZA_PRIMITIVES - 1, len(prims)
    remaining = SPONZA_TRIANGLES - tri
    assert remaining > 3, remaining
    nq = (remaining - 1) // 2                                     # quads; the count is odd, so one extra triangle closes it
    nu = max(1, int(math.sqrt(nq)))
    nv = nq // nu
    strip = [_quad((lo[0] + 40, fy + 2, col_z[0] + 60), (0, 0, (col_z[1] - col_z[0]) - 120), (ext[0] - 80, 0, 0), nu, nv)]
    rest = nq - nu * nv
    if rest:
        strip.append(_quad((lo[0] + 40, fy + 4, col_z[0] + 60), (0, 0, 50), (ext[0] - 80, 0, 0), 1, rest))
    pos, uv, nrm, idx = _merge(strip)
    odd = remaining - 2 * nq
    if odd:
        b = len(pos)
        extra_p = np.array([[lo[0] + 40, fy + 6, col_z[0] + 60], [lo[0] + 40, fy + 6, col_z[0] + 110], [lo[0] + 90, fy + 6, col_z[0] + 60]], np.float32)
        pos = np.concatenate([pos, extra_p]); uv = np.concatenate([uv, np.array([[0, 0], [1, 0], [0, 1]], np.float32)])
        nrm = np.concatenate([nrm, np.array([[0, 1, 0]] * 3, np.float32)]); idx = np.concatenate([idx, np.array([[b, b + 1, b + 2]], np.uint32)])
    tri += _add_part(d, (pos, uv, nrm, idx), opaque[16], prims)
    assert tri == SPONZA_TRIANGLES and len(prims) == SPONZA_PRIMITIVES, (tri, len(prims))
    mesh = d.add_mesh(prims)
    S = SPONZA_NODE_SCALE
    xf = np.diag([S, S, S, 1.0]).astype(np.float32)
    d.add_instance(mesh, xf)
    # ---- light: one 2-triangle quad just under the roof opening, facing -Y, its own instance, EmissionMode::OVERRIDE.
    # The material itself is emissive so that FindEmissives counts it (WaveFrontRenderer.cpp:456-464 tests that count).
    lq = _quad((ox0 + 0.1 * (ox1 - ox0), ry - 40, oz0 + 0.1 * (oz1 - oz0)), (0.8 * (ox1 - ox0), 0, 0), (0, 0, 0.8 * (oz1 - oz0)))
    lprims = []
    _add_part(d, lq, light_mat, lprims)
    d.add_instance(d.add_mesh(lprims), xf, emission_mode=2, override_radiance=light_radiance, scale=light_scale)
    # ---- C3: `extra_lights` quads of 0.2 x 0.2 m, positions uniform in the AABB shrunk by 10 %, normals -Y,
    # radiance uniform in [5,50]^3, RNG = xorshift32 seeded WangHash(1024) = 0x...
    if extra_lights:
        r2 = _Xorshift(_wang_hash(1024))
        wlo, whi = lo * S, hi * S
        c, e = (wlo + whi) / 2, (whi - wlo) * 0.9
        for _ in range(extra_lights):
            p = np.array([c[0] + (r2.f() - 0.5) * e[0], c[1] + (r2.f() - 0.5) * e[1], c[2] + (r2.f() - 0.5) * e[2]])
            rad = (r2.uniform(5, 50), r2.uniform(5, 50), r2.uniform(5, 50))
            q = _quad(p - np.array([0.1, 0, 0.1]), (0.2, 0, 0), (0, 0, 0.2))
            qp = []
            _add_part(d, q, light_mat, qp)
            d.add_instance(d.add_mesh(qp), None, emission_mode=2, override_radiance=rad, scale=1.0)
    # ---- camera: AABB centre - 0.35 extent along +X, 0.15 height above the floor, looking +X, up +Y
    wlo, whi = lo * S, hi * S
    centre, wext = (wlo + whi) / 2, whi - wlo
    eye = (centre[0] - 0.35 * wext[0], wlo[1] + 0.15 * wext[1], centre[2])
    fwd, up = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    right = np.cross(up, fwd)                                     # rotation-matrix columns: right x up = forward (Camera.cpp:128-140)
    d.set_camera(eye, tuple(right), tuple(up), tuple(fwd), 90.0)
    return d


def sandbox_camera_pose(desc, k):
    """Camera of TraceFrame k of the `sandbox` workload (the reference's own default setting with a moving camera: Sandbox/src/Application.cpp:89-93,
    OutputLayer.cpp:492-495): the scene's camera walking 0.4 % of the view distance per frame forward and sideways while it yaws 0.6 degrees and nods a
    little.  Returns (position, right, up, forward, fov) as float32 arrays — the arguments of SetCamera."""
    c = desc.camera
    pos0 = np.asarray(c["position"], np.float64); fwd0 = np.asarray(c["forward"], np.float64); up0 = np.asarray(c["up"], np.float64); right0 = np.asarray(c["right"], np.float64)
    step = 0.12                                                    # world units per frame (the stand-in's atrium is ~ 30 long after its 0.008 node scale)
    yaw, nod = 0.0105 * k, 0.02 * math.sin(0.35 * k)
    fwd = fwd0 * math.cos(yaw) + right0 * math.sin(yaw) + up0 * nod
    fwd /= np.linalg.norm(fwd)
    right = np.cross(up0, fwd); right /= np.linalg.norm(right)    # rotation-matrix columns as the scene's own camera builds them: right = up x forward
    up = np.cross(fwd, right)
    pos = pos0 + fwd0 * (step * k) + right0 * (0.05 * k)
    return (pos.astype(np.float32), right.astype(np.float32), up.astype(np.float32), fwd.astype(np.float32), float(c["fov"]))


def _wang_hash(s):
    s = ((s ^ 61) ^ (s >> 16)) & 0xFFFFFFFF
    s = (s * 9) & 0xFFFFFFFF
    s = s ^ (s >> 4)
    s = (s * 0x27d4eb2d) & 0xFFFFFFFF
    s = s ^ (s >> 15)
    return s


def foliage_stress(copies=10000, tris_per_plant=1000):
    """C5: deep-BVH stress, ``copies`` x ``tris_per_plant`` triangles flattened into one primitive set + a sky quad light."""
    rng = _Xorshift(_wang_hash(5))
    d = SceneDescription()
    leaf = d.add_material(diffuse_color=(0.25, 0.5, 0.2, 1.0), metallic_factor=0.0, roughness_factor=0.7)
    ground = d.add_material(diffuse_color=(0.4, 0.35, 0.3, 1.0), metallic_factor=0.0, roughness_factor=0.9)
    light_mat = d.add_material(emission=(1, 1, 1), metallic_factor=0.0)
    nq = tris_per_plant // 2
    # one plant: nq leaf quads around a stem
    base_p, base_n, base_uv = [], [], []
    for _ in range(nq):
        a, tilt, r, h = rng.uniform(0, 6.28), rng.uniform(-0.4, 1.0), rng.uniform(0.1, 0.5), rng.uniform(0.1, 2.0)
        eu = np.array([math.cos(a) * r, math.sin(tilt) * r, math.sin(a) * r]); ev = np.cross(eu, (0, 1, 0)); ev = ev / np.linalg.norm(ev) * r * 0.4
        p0 = np.array([math.cos(a) * 0.05, h, math.sin(a) * 0.05])
        n = np.cross(eu, ev); n /= np.linalg.norm(n)
        base_p.append([p0, p0 + eu, p0 + eu + ev, p0 + ev]); base_n.append([n] * 4); base_uv.append([[0, 0], [1, 0], [1, 1], [0, 1]])
    base_p, base_n, base_uv = np.asarray(base_p, np.float32), np.asarray(base_n, np.float32), np.asarray(base_uv, np.float32)
    quad_idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    prims = []
    per_prim = 64                                                  # plants per primitive (keeps 32-bit index buffers modest)
    side = int(math.ceil(math.sqrt(copies)))
    for start in range(0, copies, per_prim):
        P, N, U = [], [], []
        for c in range(start, min(copies, start + per_prim)):
            gx, gz = c % side, c // side
            ang, sc = rng.uniform(0, 6.28), rng.uniform(0.7, 1.4)
            ca, sa = math.cos(ang), math.sin(ang)
            R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], np.float32)
            off = np.array([gx * 1.2 + rng.uniform(-0.3, 0.3), 0.0, gz * 1.2 + rng.uniform(-0.3, 0.3)], np.float32)
            P.append((base_p.reshape(-1, 3) @ R.T) * sc + off); N.append(base_n.reshape(-1, 3) @ R.T); U.append(base_uv.reshape(-1, 2))
        P, N, U = np.concatenate(P), np.concatenate(N), np.concatenate(U)
        nquads = len(P) // 4
        idx = (np.arange(nquads, dtype=np.uint32)[:, None] * 4 + quad_idx[None, :]).reshape(-1, 3)
        _add_part(d, (P, U, N, idx), leaf, prims)
    L = side * 1.2
    _add_part(d, _quad((-5, 0, -5), (0, 0, L + 10), (L + 10, 0, 0), 64, 64), ground, prims)
    d.add_instance(d.add_mesh(prims))
    lp = []
    _add_part(d, _quad((0, 12, 0), (L, 0, 0), (0, 0, L)), light_mat, lp)
    d.add_instance(d.add_mesh(lp), None, emission_mode=2, override_radiance=(1.0, 0.95, 0.9), scale=4.0)
    fwd = np.array([1.0, -0.25, 1.0]); fwd /= np.linalg.norm(fwd)
    up0 = np.array([0.0, 1.0, 0.0]); right = np.cross(up0, fwd); right /= np.linalg.norm(right); up = np.cross(fwd, right)
    d.set_camera((-2.0, 4.0, -2.0), tuple(right), tuple(up), tuple(fwd), 90.0)
    return d


# ---------------------------------------------------------------------------------------------------------------------
# flat scene file for callers of the C ABI that are not Python (examples/render_scene.c)
# ---------------------------------------------------------------------------------------------------------------------
SCENE_FILE_MAGIC = 0x314D4C53          # "SLM1"
_MATERIAL_TEXTURES = ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
                      "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture")
_MATERIAL_SCALARS = ("transmission_factor", "clearcoat_factor", "clearcoat_roughness_factor", "index_of_refraction", "specular_factor",
                     "specular_tint_factor", "subsurface_factor", "luminance", "anisotropic", "sheen_factor", "sheen_tint_factor",
                     "metallic_factor", "roughness_factor")


def write_scene_file(desc, path):
    """Little-endian dump of a SceneDescription, in the order a caller replays it through the factories of include/lumen_mi.h:
    u32 magic; camera 13 f32 (position, right, up, forward, fov);
    u32 nTex   { u32 w, h, srgb; w*h*4 bytes RGBA8 };
    u32 nMat   { f32 diffuse[4], emission[3]; u32 texture index x 8; f32 x 13 scalars (order of lumen_mi_material_data); f32 tint[3], transmittance[3] };
    u32 nPrim  { u32 material, nVertices, nIndices; nVertices * 48 bytes interleaved vertices; nIndices u32 };
    u32 nMesh  { u32 nPrims; u32 primitive index x nPrims };
    u32 nInst  { u32 mesh; f32 transform[16] row-major; i32 emission mode; f32 radiance[3], scale; i32 override material (-1 = none) }."""
    import struct
    with open(path, "wb") as f:
        c = desc.camera
        f.write(struct.pack("<I13f", SCENE_FILE_MAGIC, *c["position"], *c["right"], *c["up"], *c["forward"], c["fov"]))
        f.write(struct.pack("<I", len(desc.textures)))
        for t in desc.textures:
            px = np.ascontiguousarray(t["pixels"], np.uint8)
            f.write(struct.pack("<3I", px.shape[1], px.shape[0], int(t["srgb"]))); f.write(px.tobytes())
        f.write(struct.pack("<I", len(desc.materials)))
        for m in desc.materials:
            f.write(struct.pack("<7f", *m["diffuse_color"], *m["emission"]))
            f.write(struct.pack("<8I", *[m[k] for k in _MATERIAL_TEXTURES]))
            f.write(struct.pack("<13f", *[m[k] for k in _MATERIAL_SCALARS]))
            f.write(struct.pack("<6f", *m["tint_factor"], *m["transmittance"]))
        f.write(struct.pack("<I", len(desc.primitives)))
        for p in desc.primitives:
            v = np.ascontiguousarray(p["vertices"], np.float32).reshape(-1, 12); i = np.ascontiguousarray(p["indices"], np.uint32).ravel()
            f.write(struct.pack("<3I", p["material"], v.shape[0], i.size)); f.write(v.tobytes()); f.write(i.tobytes())
        f.write(struct.pack("<I", len(desc.meshes)))
        for m in desc.meshes:
            f.write(struct.pack("<I", len(m))); f.write(struct.pack("<%dI" % len(m), *m))
        f.write(struct.pack("<I", len(desc.instances)))
        for inst in desc.instances:
            f.write(struct.pack("<I16f", inst["mesh"], *np.asarray(inst["transform"], np.float32).ravel()))
            f.write(struct.pack("<i4fi", inst["emission_mode"], *inst["override_radiance"], inst["scale"], inst["override_material"]))


def read_scene_file(path):
    """Inverse of write_scene_file (the C reader is examples/render_scene.c)."""
    import struct
    data = open(path, "rb").read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, data, pos); pos += struct.calcsize("<" + fmt)
        return v

    magic, = take("I")
    if magic != SCENE_FILE_MAGIC:
        raise ValueError("not a scene file")
    d = SceneDescription()
    d.textures, d.materials = [], []
    cam = take("13f")
    d.set_camera(cam[0:3], cam[3:6], cam[6:9], cam[9:12], cam[12])
    for _ in range(take("I")[0]):
        w, h, srgb = take("3I")
        px = np.frombuffer(data, np.uint8, w * h * 4, pos).reshape(h, w, 4).copy(); pos += w * h * 4
        d.textures.append(dict(pixels=px, srgb=bool(srgb)))
    for _ in range(take("I")[0]):
        v = take("7f"); t = take("8I"); sc = take("13f"); tt = take("6f")
        m = dict(diffuse_color=tuple(v[0:4]), emission=tuple(v[4:7]), tint_factor=tuple(tt[0:3]), transmittance=tuple(tt[3:6]))
        m.update(zip(_MATERIAL_TEXTURES, t)); m.update(zip(_MATERIAL_SCALARS, sc))
        d.materials.append(m)
    for _ in range(take("I")[0]):
        mat, nv, ni = take("3I")
        v = np.frombuffer(data, np.float32, nv * 12, pos).reshape(nv, 12).copy(); pos += nv * 48
        i = np.frombuffer(data, np.uint32, ni, pos).copy(); pos += ni * 4
        d.primitives.append(dict(vertices=v, indices=i, material=mat, index_size=4))
    for _ in range(take("I")[0]):
        n, = take("I")
        d.meshes.append(list(take("%dI" % n)))
    for _ in range(take("I")[0]):
        v = take("I16f"); w = take("i4fi")
        d.instances.append(dict(mesh=v[0], transform=np.float32(v[1:17]).reshape(4, 4), emission_mode=w[0], override_radiance=tuple(w[1:4]),
                                scale=float(w[4]), override_material=w[5]))
    if pos != len(data):
        raise ValueError("trailing bytes in scene file")
    return d
