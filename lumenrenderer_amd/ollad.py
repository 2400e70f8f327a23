"""Reader / writer of the reference's ``.ollad`` scene cache.

The reference never feeds glTF to the renderer directly: ``LumenPTModelConverter::ConvertGLTF`` rewrites it once into an
``.ollad`` file and ``LoadFile`` reads that (LumenPT/src/Tools/LumenPTModelConverter.cpp:24-70 convert, :72-273 load, :275-316
nodes, :336-560 content, :563-621 header / output; structs LumenPTModelConverter.h:62-209).  Layout, little endian, no padding:

    u64 headerSize
    header:  u64 nTex   { u64 blobOffset, u64 byteSize, u64 textureType } x nTex        (one per glTF IMAGE)
             u64 nMat   HeaderMaterial x nMat (136 bytes: colour 4f, emission 3f, 8 x i32 image ids (-1 = default texture),
                        13 scalar factors, tint 3f, transmittance 3f)
             u64 nMesh  { u32 nPrim, { u64 vtxOffset, u64 vtxBytes, u64 idxOffset, u64 idxBytes, u32 indexSize, u32 materialId } x nPrim } x nMesh
             u64 nScene { u32 nRootNodes, u32 nameLength, name, nodes depth-first:
                          { u32 nameLength, u32 nChildren, f32[16] local matrix (column-major), i32 meshId (-1 = none) }, name, children ... }
    blob:    the image files as they are (PNG / JPEG, decoded at load time), interleaved 48-byte vertices
             (position 3f, uv 2f, normal 3f, tangent 4f), raw indices (16 or 32 bit)

No ``.ollad`` file ships with the reference, so the byte layout is pinned by its reader / writer source only; the tests
round-trip the reference's sample glTF assets (glTF -> .ollad -> scene) against the direct glTF ingest of ``gltf.py``.
"""
import base64
import os
import struct
from urllib.parse import unquote

import numpy as np

from . import gltf as _gltf
from .scenes import SceneDescription, generate_tangents, interleave

# TextureType (LumenPTModelConverter.h:66-77)
T_UNSPECIFIED, T_DIFFUSE, T_NORMAL, T_EMISSIVE, T_METAL_ROUGHNESS, T_TRANSMISSIVE, T_CLEARCOAT, T_CLEARCOAT_ROUGHNESS, T_TINT = range(9)

_MAT = struct.Struct("<4f3f8i13f3f3f")          # HeaderMaterial, 136 bytes
_PRIM = struct.Struct("<4Q2I")                  # HeaderPrimitive, 40 bytes
_NODE = struct.Struct("<2I16fi")                # HeaderNode::m_Header, 76 bytes
assert _MAT.size == 136 and _PRIM.size == 40 and _NODE.size == 76


def _image_bytes(doc, buffers, base, img):
    if "uri" in img and img["uri"].startswith("data:"):
        return base64.b64decode(img["uri"].split(",", 1)[1])
    if "uri" in img:
        with open(os.path.join(base, unquote(img["uri"])), "rb") as f:
            return f.read()
    bv = doc["bufferViews"][img["bufferView"]]
    return bytes(buffers[bv["buffer"]][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]])


def write_ollad(gltf_path, dst_path):
    """glTF (.gltf / .glb) -> .ollad, following GenerateContent / GenerateHeader / OutputToFile."""
    doc, glb_blob = _gltf._read_container(gltf_path)
    base = os.path.dirname(gltf_path)
    buffers = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            buffers.append(glb_blob)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(os.path.join(base, unquote(uri)), "rb") as f:
                buffers.append(f.read())
    blob = bytearray()

    def put(data):
        off = len(blob)
        blob.extend(data)
        return off, len(data)

    textures = []                                   # [offset, size, type] per image (TextureToBlob :623-662)
    for img in doc.get("images", []):
        off, size = put(_image_bytes(doc, buffers, base, img))
        textures.append([off, size, T_UNSPECIFIED])

    def image_of(info, ttype):
        if not info or info.get("index", -1) < 0:
            return -1
        src = doc["textures"][info["index"]]["source"]
        textures[src][2] = ttype                    # one type per image: the last material to reference it wins (:361-398)
        return src

    materials = []
    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})
        color = list(pbr.get("baseColorFactor", (1, 1, 1, 1)))
        emission = list(m.get("emissiveFactor", (0, 0, 0)))
        ids = [image_of(pbr.get("baseColorTexture"), T_DIFFUSE), image_of(m.get("normalTexture"), T_NORMAL),
               image_of(pbr.get("metallicRoughnessTexture"), T_METAL_ROUGHNESS), image_of(m.get("emissiveTexture"), T_EMISSIVE),
               -1, -1, -1, -1]                      # transmission / clear coat / clear-coat roughness / tint textures: not ingested
        tr = ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0) if "KHR_materials_transmission" in ext else 0.0
        sheen, sheen_tint = (ext["KHR_materials_sheen"].get("sheenRoughnessFactor", 0.0), 1.0) if "KHR_materials_sheen" in ext else (0.0, 0.0)
        ior = ext["KHR_materials_ior"].get("ior", 1.0) if "KHR_materials_ior" in ext else 1.0
        cc = ext.get("KHR_materials_clearcoat")
        cc_f, cc_r = (cc.get("clearcoatFactor", 0.0), cc.get("clearcoatRoughnessFactor", 0.0)) if cc is not None else (0.0, 0.0)
        spec, spec_tint = (ext["KHR_materials_specular"].get("specularFactor", 0.0), 1.0) if "KHR_materials_specular" in ext else (0.0, 0.0)
        # scalar order of HeaderMaterial: transmission, clearCoat, clearCoatRoughness, ior, specular, specularTint, subSurface,
        # luminance, anisotropic, sheen, sheenTint, metallic, roughness
        scalars = [tr, cc_f, cc_r, ior, spec, spec_tint, 0.0, 1.0, 0.0, sheen, sheen_tint,
                   pbr.get("metallicFactor", 1.0), max(0.01, pbr.get("roughnessFactor", 1.0))]
        materials.append(_MAT.pack(*color, *emission, *ids, *scalars, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))

    meshes = []
    for mesh in doc.get("meshes", []):
        prims = []
        for p in mesh["primitives"]:
            at = p["attributes"]
            pos = _gltf._accessor(doc, buffers, at["POSITION"]).astype(np.float32)
            nrm = _gltf._accessor(doc, buffers, at["NORMAL"]).astype(np.float32) if "NORMAL" in at else None
            uv = _gltf._accessor(doc, buffers, at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
            if "indices" in p:
                raw = _gltf._accessor(doc, buffers, p["indices"])
                index_size = raw.dtype.itemsize if raw.dtype.itemsize in (2, 4) else 4
                idx = raw.astype(np.uint32).ravel()
            else:
                idx, index_size = np.arange(len(pos), dtype=np.uint32), 4
            idx = idx[: 3 * (len(idx) // 3)]
            if "TANGENT" in at:
                tang = _gltf._accessor(doc, buffers, at["TANGENT"]).astype(np.float32)
            else:
                tang = generate_tangents(pos, nrm if nrm is not None else np.tile(np.float32([0, 1, 0]), (len(pos), 1)), uv, idx)
            v_off, v_size = put(interleave(pos, uv, nrm, tang).astype("<f4").tobytes())
            i_off, i_size = put(idx.astype("<u2" if index_size == 2 else "<u4").tobytes())
            prims.append(_PRIM.pack(v_off, v_size, i_off, i_size, index_size, p.get("material", 0)))
        meshes.append(prims)

    def node_bytes(ni):
        node = doc["nodes"][ni]
        name = node.get("name", "").encode("utf-8")
        local = _gltf._node_local(node).astype(np.float32)
        children = node.get("children", [])
        out = _NODE.pack(len(name), len(children), *local.T.ravel().tolist(), node.get("mesh", -1)) + name
        for c in children:
            out += node_bytes(c)
        return out

    header = bytearray()
    header += struct.pack("<Q", len(textures))
    for off, size, ttype in textures:
        header += struct.pack("<3Q", off, size, ttype)
    header += struct.pack("<Q", len(materials)) + b"".join(materials)
    header += struct.pack("<Q", len(meshes))
    for prims in meshes:
        header += struct.pack("<I", len(prims)) + b"".join(prims)
    scenes = doc.get("scenes") or [{"nodes": list(range(len(doc.get("nodes", []))))}]
    header += struct.pack("<Q", len(scenes))
    for sc in scenes:
        name = sc.get("name", "").encode("utf-8")
        roots = sc.get("nodes", [])
        header += struct.pack("<2I", len(roots), len(name)) + name
        for n in roots:
            header += node_bytes(n)
    with open(dst_path, "wb") as f:
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(blob)
    return dst_path


def read_ollad(path, image_loader=None, scene=0):
    """.ollad -> SceneDescription, following LoadFile / LoadNode (one mesh instance per node with a mesh, EmissionMode ENABLED)."""
    if image_loader is None:
        image_loader = _gltf._pil_loader
    with open(path, "rb") as f:
        data = f.read()
    (header_size,) = struct.unpack_from("<Q", data, 0)
    blob = memoryview(data)[8 + header_size:]
    pos = 8

    def take(fmt):
        nonlocal pos
        s = struct.Struct(fmt)
        out = s.unpack_from(data, pos)
        pos += s.size
        return out

    d = SceneDescription()
    (n_tex,) = take("<Q")
    tex = []
    for _ in range(n_tex):
        off, size, ttype = take("<3Q")
        px = np.array(image_loader(bytes(blob[off: off + size])), np.uint8)
        if ttype == T_METAL_ROUGHNESS:
            px[..., 1] = np.maximum(px[..., 1], 1)                           # roughness >= 1/255 (:121-128)
        tex.append(d.add_texture(px, ttype in (T_DIFFUSE, T_EMISSIVE)))       # sRGB decode for base colour and emissive only (:130-133)
    (n_mat,) = take("<Q")
    mats = []
    for _ in range(n_mat):
        v = _MAT.unpack_from(data, pos)
        pos += _MAT.size
        color, emission, ids, sc, tint, transmittance = v[0:4], v[4:7], v[7:15], v[15:28], v[28:31], v[31:34]
        kw = dict(diffuse_color=color, emission=emission, tint_factor=tint, transmittance=transmittance,
                  transmission_factor=sc[0], clearcoat_factor=sc[1], clearcoat_roughness_factor=sc[2], index_of_refraction=sc[3],
                  specular_factor=sc[4], specular_tint_factor=sc[5], subsurface_factor=sc[6], luminance=sc[7], anisotropic=sc[8],
                  sheen_factor=sc[9], sheen_tint_factor=sc[10], metallic_factor=sc[11], roughness_factor=sc[12])
        for field, i in zip(("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
                             "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture"), ids):
            if i != -1:
                kw[field] = tex[i]                                            # else: the converter's default textures = SceneDescription's
        mats.append(d.add_material(**kw))
    (n_mesh,) = take("<Q")
    meshes = []
    for _ in range(n_mesh):
        (n_prim,) = take("<I")
        prims = []
        for _ in range(n_prim):
            v_off, v_size, i_off, i_size, index_size, material = take("<4Q2I")
            verts = np.frombuffer(blob[v_off: v_off + v_size], "<f4").reshape(-1, 12).copy()
            idx = np.frombuffer(blob[i_off: i_off + i_size], "<u2" if index_size == 2 else "<u4").astype(np.uint32)
            prims.append(d.add_primitive(verts, idx, mats[material], index_size))
        meshes.append(d.add_mesh(prims))
    (n_scene,) = take("<Q")

    def node(parent, keep):
        name_len, n_children, *rest = take("<2I16fi")
        nonlocal pos
        pos += name_len
        local = np.asarray(rest[:16], np.float32).reshape(4, 4).T            # glm::make_mat4: column-major
        mesh_id = rest[16]
        world = _gltf.compose(parent, local)
        if keep and mesh_id != -1:
            d.add_instance(meshes[mesh_id], world)
        for _ in range(n_children):
            node(world, keep)

    for s in range(n_scene):
        n_roots, name_len = take("<2I")
        pos += name_len
        for _ in range(n_roots):
            node(np.eye(4, dtype=np.float32), s == scene)
    return d
