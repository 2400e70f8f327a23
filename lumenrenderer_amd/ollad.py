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
    blob:    the image files as they are (PNG / JPEG, decoded at load time), interleaved 64-BYTE vertices, raw indices (16 or 32 bit)

The vertex record is the reference's ``Vertex`` (Shaders/CppCommon/ModelStructs.h:21-28) as its compilers lay it out.  ``LUMEN`` is defined nowhere in
the reference's build, so the members are CUDA's vector types, whose float2 is 8-byte and float4 16-byte aligned: position 3f at byte 0, uv 2f at 16,
normal 3f at 24, tangent 4f at 48, zero padding in between — 64 bytes, not the 48 a packed struct would take (rounds 1-3 wrote 48).  Pinned in round 4 by
running the reference's own converter: tests/test_cpu_host.py builds SceneManager.cpp + LumenPTModelConverter.cpp from the mounted tree, lets them
convert CornellBox/scene.gltf, and holds this module's writer to the file they wrote, byte for byte.
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

VERTEX_STRIDE = 64                               # sizeof(Vertex) under CUDA's alignment rules (see the module docstring)


def pack_vertices(v12):
    """[n][12] floats (position uv normal tangent, the 48-byte order every other module uses) -> the file's 64-byte records"""
    v12 = np.ascontiguousarray(v12, "<f4").reshape(-1, 12)
    out = np.zeros((v12.shape[0], 16), "<f4")
    out[:, 0:3] = v12[:, 0:3]; out[:, 4:6] = v12[:, 3:5]; out[:, 6:9] = v12[:, 5:8]; out[:, 12:16] = v12[:, 8:12]
    return out.tobytes()


def unpack_vertices(raw):
    if len(raw) % VERTEX_STRIDE:
        raise ValueError(f"vertex buffer of {len(raw)} bytes is not a whole number of {VERTEX_STRIDE}-byte vertices")
    v = np.frombuffer(raw, "<f4").reshape(-1, 16)
    return np.ascontiguousarray(np.concatenate([v[:, 0:3], v[:, 4:6], v[:, 6:9], v[:, 12:16]], axis=1), np.float32)


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
            v_off, v_size = put(pack_vertices(interleave(pos, uv, nrm, tang)))
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
            verts = unpack_vertices(bytes(blob[v_off: v_off + v_size]))
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
        if mesh_id != -1:
            if keep:
                d.add_instance(meshes[mesh_id], world)
            world = local                                                    # reference quirk: children of a mesh node inherit its local matrix only (gltf.py walk)
        for _ in range(n_children):
            node(world, keep)

    for s in range(n_scene):
        n_roots, name_len = take("<2I")
        pos += name_len
        for _ in range(n_roots):
            node(np.eye(4, dtype=np.float32), s == scene)
    return d


def _png_rgba8(px):
    """RGBA8 image -> PNG bytes (8-bit RGBA, filter 0 on every row, one IDAT): what an .ollad blob holds per image."""
    import zlib
    px = np.ascontiguousarray(px, np.uint8)
    h, w = px.shape[:2]

    def chunk(kind, data):
        body = kind + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def write_ollad_from_description(desc, dst_path):
    """SceneDescription -> .ollad (the layout above): every texture a material uses as a PNG image typed by its use, the materials, one mesh per
    description mesh, and ONE scene whose root nodes are the mesh instances (local matrix = the instance's world matrix).  What the format cannot say is
    refused: emission overrides / override materials of an instance (LoadNode creates ENABLED instances, LumenPTModelConverter.cpp:275-316), and an sRGB flag
    that disagrees with the texture's use (LoadFile derives it from the type: base colour and emissive only, :130-133)."""
    defaults = {desc.tex_white, desc.tex_normal, desc.tex_metal_rough, desc.tex_emissive}
    slots = (("diffuse_texture", T_DIFFUSE), ("normal_map", T_NORMAL), ("metallic_roughness_texture", T_METAL_ROUGHNESS), ("emissive_texture", T_EMISSIVE),
             ("transmission_texture", T_TRANSMISSIVE), ("clearcoat_texture", T_CLEARCOAT), ("clearcoat_roughness_texture", T_CLEARCOAT_ROUGHNESS), ("tint_texture", T_TINT))
    blob = bytearray()

    def put(data):
        off = len(blob)
        blob.extend(data)
        return off, len(data)

    image_of, textures = {}, []                     # description texture index -> image id; [offset, size, type] per image
    materials = []
    for m in desc.materials:
        ids = []
        for field, ttype in slots:
            t = m[field]
            if t in defaults:
                ids.append(-1)
                continue
            if t not in image_of:
                off, size = put(_png_rgba8(desc.textures[t]["pixels"]))
                image_of[t] = len(textures); textures.append([off, size, ttype])
            if textures[image_of[t]][2] != ttype:
                raise ValueError("a texture used in two different slots cannot be typed in an .ollad file")
            if desc.textures[t]["srgb"] != (ttype in (T_DIFFUSE, T_EMISSIVE)):
                raise ValueError(f"texture {t}: sRGB flag disagrees with its use as {field}")
            ids.append(image_of[t])
        scalars = [m[k] for k in ("transmission_factor", "clearcoat_factor", "clearcoat_roughness_factor", "index_of_refraction", "specular_factor", "specular_tint_factor",
                                  "subsurface_factor", "luminance", "anisotropic", "sheen_factor", "sheen_tint_factor", "metallic_factor", "roughness_factor")]
        materials.append(_MAT.pack(*m["diffuse_color"], *m["emission"], *ids, *scalars, *m["tint_factor"], *m["transmittance"]))
    meshes = []
    for prims in desc.meshes:
        out = []
        for pi in prims:
            p = desc.primitives[pi]
            index_size = p["index_size"]
            v_off, v_size = put(pack_vertices(p["vertices"]))
            i_off, i_size = put(p["indices"].astype("<u2" if index_size == 2 else "<u4").tobytes())
            out.append(_PRIM.pack(v_off, v_size, i_off, i_size, index_size, p["material"]))
        meshes.append(out)
    header = bytearray()
    header += struct.pack("<Q", len(textures))
    for off, size, ttype in textures:
        header += struct.pack("<3Q", off, size, ttype)
    header += struct.pack("<Q", len(materials)) + b"".join(materials)
    header += struct.pack("<Q", len(meshes))
    for prims in meshes:
        header += struct.pack("<I", len(prims)) + b"".join(prims)
    name = b"scene"
    header += struct.pack("<Q", 1) + struct.pack("<2I", len(desc.instances), len(name)) + name
    for k, inst in enumerate(desc.instances):
        if inst["emission_mode"] != 0 or inst["override_material"] >= 0:
            raise ValueError("an .ollad node cannot carry an emission override or an override material")
        nm = f"instance{k}".encode()
        header += _NODE.pack(len(nm), 0, *np.asarray(inst["transform"], np.float32).T.ravel().tolist(), inst["mesh"]) + nm
    with open(dst_path, "wb") as f:
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(blob)
    return dst_path
