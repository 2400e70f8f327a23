"""Rows of tests/golden/ref_kat6.npz — what the reference's scene-facing kernel bodies (ExtractSurfaceDataGpu, GenerateMotionVector, FindEmissivesGpu,
BuildLightDataBufferGPU) computed on a small scene of 1 x 1 textures (generator oracle/ref_kat/gen_kat6.cpp) — as a SceneDescription both the oracle and the
product load through their ordinary scene API, plus the hit / ray rows and the reference outputs."""
import os
import numpy as np

W, H = 64, 48
N = W * H
_G = None


def gold():
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_kat6.npz"))
    return _G


def f32(words):
    return np.ascontiguousarray(words, dtype=np.int64).astype(np.uint32).view(np.float32)


def scene():
    """SceneDescription of the rows: textures (1 x 1, none sRGB-flagged), materials, one mesh per primitive, instances in table order (entry i = instance i)."""
    from lumenrenderer_amd.scenes import SceneDescription
    g = gold()
    d = SceneDescription()
    tex = [d.add_texture(np.array([[row[1:5]]], np.uint8), False) for row in g["xtex"]]
    slots = ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture", "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture")
    mats = []
    for row in g["xmat"]:
        v = f32(row[1:27]); ids = row[27:35]
        p = v[15:26]               # metallic subsurface specular roughness spectint anisotropic sheen sheentint clearcoat clearcoatgloss transmission
        kw = dict(diffuse_color=tuple(v[0:4]), emission=tuple(v[4:7]), tint_factor=tuple(v[7:10]), luminance=float(v[10]), transmittance=tuple(v[11:14]), index_of_refraction=float(v[14]),
                  metallic_factor=float(p[0]), subsurface_factor=float(p[1]), specular_factor=float(p[2]), roughness_factor=float(p[3]), specular_tint_factor=float(p[4]), anisotropic=float(p[5]),
                  sheen_factor=float(p[6]), sheen_tint_factor=float(p[7]), clearcoat_factor=float(p[8]), clearcoat_roughness_factor=float(np.float32(1.0) - p[9]), transmission_factor=float(p[10]))
        for name, t in zip(slots, ids):
            kw[name] = tex[int(t)]
        mats.append(d.add_material(**kw))
    meshes = []
    for prow in g["xprim"]:
        p = int(prow[0])
        verts = f32(g["xvert"][g["xvert"][:, 0] == p][:, 2:]).reshape(-1, 12)
        idx = g["xidx"][g["xidx"][:, 0] == p][:, 1:].astype(np.uint32).ravel()
        meshes.append(d.add_mesh([d.add_primitive(verts, idx, mats[int(prow[1])], 4)]))
    for row in g["xinst"]:
        v = f32(row[3:23])
        d.add_instance(meshes[int(row[1])], v[:16].reshape(4, 4), emission_mode=int(row[2]), override_radiance=tuple(v[16:19]), scale=float(v[19]))
    return d


def hits(which):
    """(hits9, rays9, want35) of ray set `which` (0: the primary wave, 1: a deeper wave)"""
    g = gold()
    h = g["xhit"]; h = h[h[:, 1] == which]
    s = g["xsurf"]; s = s[s[:, 1] == which]
    assert np.array_equal(h[:, 0], np.arange(N)) and np.array_equal(s[:, 0], np.arange(N))
    hits9 = np.zeros((N, 9), np.uint32); hits9[:, :7] = h[:, 2:9]
    rays9 = np.ascontiguousarray(h[:, 9:18], dtype=np.uint32)
    return hits9, rays9, np.ascontiguousarray(s[:, 2:], dtype=np.uint32)


def motion():
    g = gold()
    return np.ascontiguousarray(g["xmvm"][0], dtype=np.uint32), np.ascontiguousarray(g["xmv"][:, 1:], dtype=np.uint32)


def lights():
    g = gold()
    return np.ascontiguousarray(g["xlight"][:, 1:], dtype=np.uint32), int(g["xnlights"][0, 0]), int(g["xnlights"][0, 1])


def resolved():
    """DIRECT channel after ResolveDirectLightHits on the primary-wave surfaces: [N][4] binary16 bit patterns (0 where the kernel wrote nothing)"""
    g = gold()
    assert np.array_equal(g["xres"][:, 0], np.arange(N))
    return np.ascontiguousarray(g["xres"][:, 1:], dtype=np.uint32)
