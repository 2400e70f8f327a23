"""Minimal glTF 2.0 reader that reproduces the reference's ingest rules (the path WaveFrontRenderer actually takes:
glTF -> .ollad -> renderer factories; LumenPT/src/Tools/LumenPTModelConverter.cpp:336-560, 664-928, 72-273):

  * attributes POSITION / NORMAL / TEXCOORD_0 / TANGENT; a missing TEXCOORD_0 leaves UV = (0,0) on every vertex
    (zero-initialised interleave buffer, :909-919), missing tangents are generated from the UVs (:734-900);
  * 16-bit indices stay 16-bit in the file and are widened by CreatePrimitive;
  * materials: baseColor/emissive/metallic factors, roughness clamped to >= 0.01 (:399), Disney extras at their
    .ollad defaults (luminance 1, transmittance 0, tint 0, ior 1, ...), KHR_materials_{transmission,sheen,ior,
    clearcoat,specular} factors; textures (sRGB decode only for base colour and emissive, :130-133) are decoded by
    ``image_loader`` (default: Pillow when importable; the reference uses stb_image), from files, data URIs or buffer views;
  * .gltf (external / embedded buffers) and .glb containers; triangle primitives only, indexed or not;
  * node hierarchy: local = T*R*S or the given matrix, world = parent*local (Transform.cpp:265-308); every node with
    a mesh becomes one mesh instance with EmissionMode::ENABLED; the glTF camera is ignored (as in the reference).
"""
import base64
import json
import os

import numpy as np

from .scenes import SceneDescription, generate_tangents, interleave

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(doc, buffers, idx):
    acc = doc["accessors"][idx]
    dt, nc = np.dtype(_COMP[acc["componentType"]]), _NCOMP[acc["type"]]
    count = acc["count"]
    if "bufferView" not in acc:                                             # no buffer view: zeros (glTF 2.0, 3.6.2.3)
        return np.zeros((count, nc), dt)
    bv = doc["bufferViews"][acc["bufferView"]]
    off = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    raw = np.frombuffer(buffers[bv["buffer"]], np.uint8)
    if count == 0:
        return np.zeros((0, nc), dt)
    rows = np.lib.stride_tricks.as_strided(raw[off:], shape=(count, dt.itemsize * nc), strides=(stride, 1))
    return np.ascontiguousarray(rows).view(dt).reshape(count, nc).copy()


def _pil_loader(src):
    """Default image decoder (PNG / JPEG -> RGBA8); the reference decodes with stb_image (LumenPTModelConverter.cpp:117)."""
    import io
    from PIL import Image
    img = Image.open(io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src)
    return np.asarray(img.convert("RGBA"), np.uint8)


def _read_container(path):
    """(json document, [binary chunk] or None) of a .gltf or a .glb file."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"glTF":
        return json.loads(data.decode("utf-8")), None
    import struct
    _, _, total = struct.unpack_from("<III", data, 0)
    off, doc, blob = 12, None, None
    while off < total:
        n, kind = struct.unpack_from("<II", data, off)
        chunk = data[off + 8: off + 8 + n]
        if kind == 0x4E4F534A: doc = json.loads(chunk.decode("utf-8"))
        elif kind == 0x004E4942 and blob is None: blob = chunk
        off += 8 + n
    return doc, blob


def _node_local(node):
    """Local matrix of a glTF node in FLOAT32 with the reference's arithmetic (LumenPTModelConverter::LoadNodeTransform, LumenPTModelConverter.cpp:994-1023 ->
    Lumen::Transform::UpdateLocalMatrix, Transform.cpp:265-280): a `matrix` is taken as given; otherwise glm::translate(I, t) * glm::mat4_cast(q), then
    glm::scale(., s), every product and sum rounded to float in glm's order.  Row-major numpy array [row][column].  Pinned in round 4 against the .ollad file the
    reference's own converter writes (tests/test_cpu_host.py): the node matrices in it are bit-identical."""
    f = np.float32
    if "matrix" in node and not np.array_equal(np.asarray(node["matrix"], np.float32), np.eye(4, dtype=np.float32).ravel()):
        return np.asarray(node["matrix"], np.float32).reshape(4, 4).T      # glTF matrices are column-major
    t = [f(v) for v in node.get("translation", (0, 0, 0))]
    x, y, z, w = [f(v) for v in node.get("rotation", (0, 0, 0, 1))]
    s = [f(v) for v in node.get("scale", (1, 1, 1))]
    one, two = f(1), f(2)
    qxx, qyy, qzz, qxz, qxy, qyz, qwx, qwy, qwz = x * x, y * y, z * z, x * z, x * y, y * z, w * x, w * y, w * z
    # glm::mat3_cast (gtc/quaternion.inl): columns of the rotation
    c0 = [one - two * (qyy + qzz), two * (qxy + qwz), two * (qxz - qwy)]
    c1 = [two * (qxy - qwz), one - two * (qxx + qzz), two * (qyz + qwx)]
    c2 = [two * (qxz + qwy), two * (qyz - qwx), one - two * (qxx + qyy)]
    rot = np.zeros((4, 4), np.float32)                                      # [column][row], glm's storage
    rot[0, :3], rot[1, :3], rot[2, :3], rot[3, 3] = c0, c1, c2, one
    tr = np.eye(4, dtype=np.float32); tr[3, :3] = t                         # glm::translate(I, t): column 3 = I0*tx + I1*ty + I2*tz + I3, exact
    m = np.zeros((4, 4), np.float32)                                        # translate * rotation: Result[c] = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] + A3*B[c][3], summed left to right
    for c in range(4):
        acc = tr[0] * rot[c, 0]
        acc = (acc + tr[1] * rot[c, 1]).astype(np.float32)
        acc = (acc + tr[2] * rot[c, 2]).astype(np.float32)
        m[c] = (acc + tr[3] * rot[c, 3]).astype(np.float32)
    m[0] = m[0] * s[0]; m[1] = m[1] * s[1]; m[2] = m[2] * s[2]              # glm::scale: the first three columns scaled
    return m.T.copy()                                                       # -> [row][column]


def compose(parent, local):
    """world = parent * local in float32 with the operation order of glm's mat4 product (the reference composes its node
    hierarchy in glm floats, Transform.cpp:282-308): C[r][c] = ((A[r][0]*B[0][c] + A[r][1]*B[1][c]) + A[r][2]*B[2][c]) + A[r][3]*B[3][c]."""
    a, b = np.asarray(parent, np.float32), np.asarray(local, np.float32)
    out = np.zeros((4, 4), np.float32)
    for r in range(4):
        for c in range(4):
            acc = np.float32(a[r, 0] * b[0, c])
            for k in (1, 2, 3):
                acc = np.float32(acc + np.float32(a[r, k] * b[k, c]))
            out[r, c] = acc
    return out


# material extensions this reader maps onto MaterialData (LumenPTModelConverter.cpp:336-560); anything else a file REQUIRES cannot be honoured
_SUPPORTED_EXTENSIONS = {"KHR_materials_transmission", "KHR_materials_sheen", "KHR_materials_ior", "KHR_materials_clearcoat", "KHR_materials_specular"}


def load_gltf(path, image_loader=None):
    doc, glb_blob = _read_container(path)
    # glTF 2.0, 3.12: a loader that does not implement an extension listed in extensionsRequired must fail.  The reference's sample set has such a file
    # (Buggy/glTF-Draco: KHR_draco_mesh_compression — its accessors have no buffer views, so reading it "as zeros" yields 532 k degenerate triangles); the reference's
    # own loader (fx-gltf, no Draco decoder) cannot show that model either.
    unsupported = sorted(set(doc.get("extensionsRequired", [])) - _SUPPORTED_EXTENSIONS)
    if unsupported:
        raise ValueError(f"{os.path.basename(path)} requires glTF extension(s) this reader does not implement: {', '.join(unsupported)}")
    base = os.path.dirname(path)
    if image_loader is None:
        try:
            import PIL  # noqa: F401
            image_loader = _pil_loader
        except ImportError:
            image_loader = None
    buffers = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            buffers.append(glb_blob)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            from urllib.parse import unquote
            with open(os.path.join(base, unquote(uri)), "rb") as fb:
                buffers.append(fb.read())
    d = SceneDescription()
    tex_cache = {}

    def texture(info, srgb, metal_rough=False):
        if info is None or info.get("index", -1) < 0:
            return None
        src = doc["textures"][info["index"]]["source"]
        key = (src, srgb)
        if key not in tex_cache:
            if image_loader is None:
                raise ValueError("glTF image needs an image_loader(path_or_bytes) -> HxWx4 uint8")
            img = doc["images"][src]
            if "uri" in img and img["uri"].startswith("data:"):
                source = base64.b64decode(img["uri"].split(",", 1)[1])
            elif "uri" in img:
                from urllib.parse import unquote
                source = os.path.join(base, unquote(img["uri"]))
            else:
                bv = doc["bufferViews"][img["bufferView"]]
                source = bytes(buffers[bv["buffer"]][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]])
            px = np.array(image_loader(source), np.uint8)
            if metal_rough:
                px[..., 1] = np.maximum(px[..., 1], 1)                       # roughness >= 1/255 (LumenPTModelConverter.cpp:121-128)
            tex_cache[key] = d.add_texture(px, srgb)
        return tex_cache[key]

    mats = []
    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})
        kw = dict(diffuse_color=tuple(pbr.get("baseColorFactor", (1, 1, 1, 1))), emission=tuple(m.get("emissiveFactor", (0, 0, 0))),
                  metallic_factor=pbr.get("metallicFactor", 1.0), roughness_factor=max(0.01, pbr.get("roughnessFactor", 1.0)))
        for field, info, srgb, mr in (("diffuse_texture", pbr.get("baseColorTexture"), True, False), ("normal_map", m.get("normalTexture"), False, False),
                                      ("metallic_roughness_texture", pbr.get("metallicRoughnessTexture"), False, True),
                                      ("emissive_texture", m.get("emissiveTexture"), True, False)):
            t = texture(info, srgb, mr)
            if t is not None:
                kw[field] = t
        if "KHR_materials_transmission" in ext:
            kw["transmission_factor"] = ext["KHR_materials_transmission"].get("transmissionFactor", 0.0)
        if "KHR_materials_sheen" in ext:
            kw["sheen_factor"] = ext["KHR_materials_sheen"].get("sheenRoughnessFactor", 0.0); kw["sheen_tint_factor"] = 1.0
        if "KHR_materials_ior" in ext:
            kw["index_of_refraction"] = ext["KHR_materials_ior"].get("ior", 1.0)
        if "KHR_materials_clearcoat" in ext:
            kw["clearcoat_factor"] = ext["KHR_materials_clearcoat"].get("clearcoatFactor", 0.0)
            kw["clearcoat_roughness_factor"] = ext["KHR_materials_clearcoat"].get("clearcoatRoughnessFactor", 0.0)
        if "KHR_materials_specular" in ext:
            kw["specular_factor"] = ext["KHR_materials_specular"].get("specularFactor", 0.0); kw["specular_tint_factor"] = 1.0
        mats.append(d.add_material(**kw))
    meshes = []
    default_material = [None]
    for mesh in doc.get("meshes", []):
        prims = []
        for p in mesh["primitives"]:
            at = p["attributes"]
            pos = _accessor(doc, buffers, at["POSITION"]).astype(np.float32)
            nrm = _accessor(doc, buffers, at["NORMAL"]).astype(np.float32) if "NORMAL" in at else None
            uv = _accessor(doc, buffers, at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
            if p.get("mode", 4) != 4:
                continue                                                    # triangles only, as the reference's converter
            if "indices" in p:
                idx = _accessor(doc, buffers, p["indices"])
                index_size = idx.dtype.itemsize if idx.dtype.itemsize in (2, 4) else 4
                idx = idx.astype(np.uint32).ravel()
            else:
                idx, index_size = np.arange(len(pos), dtype=np.uint32), 4
            idx = idx[: 3 * (len(idx) // 3)]
            if "TANGENT" in at:
                tang = _accessor(doc, buffers, at["TANGENT"]).astype(np.float32)
            else:
                tang = generate_tangents(pos, nrm if nrm is not None else np.tile(np.float32([0, 1, 0]), (len(pos), 1)), uv, idx)
            if "material" in p:
                material = mats[p["material"]]
            else:
                if default_material[0] is None:
                    default_material[0] = d.add_material()                  # glTF default material: white, metallic 1, roughness 1
                material = default_material[0]
            prims.append(d.add_primitive(interleave(pos, uv, nrm, tang), idx, material, index_size))
        meshes.append(d.add_mesh(prims))

    def walk(ni, parent):
        node = doc["nodes"][ni]
        local = _node_local(node).astype(np.float32)                       # local matrices are stored as floats (.ollad node header)
        world = compose(parent, local)
        if "mesh" in node:
            d.add_instance(meshes[node["mesh"]], world)
            # Reference quirk (LumenPTModelConverter::LoadNode, LumenPTModelConverter.cpp:275-316): for a node WITH a mesh only the mesh instance's transform is
            # attached to the parent's (:296-297); the node's own transform, which its children attach to, stays without a parent.  Children of a mesh node
            # therefore inherit that node's LOCAL matrix only — the ancestors above it drop out (the wheels of the Cesium milk truck lose the root's Y-up
            # correction).  Reproduced; pinned by running the reference's own loader (tests/test_cpu_host.py).
            world = local
        for c in node.get("children", []):
            walk(c, world)

    scenes = doc.get("scenes") or [{"nodes": list(range(len(doc.get("nodes", []))))}]
    for n in scenes[doc.get("scene", 0)].get("nodes", []):
        walk(n, np.eye(4, dtype=np.float32))
    return d
