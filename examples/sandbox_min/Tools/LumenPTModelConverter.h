// LumenPTModelConverter.h (minimal interface tree) — a from-scratch reader of the reference's `.ollad` scene cache with the public shape of the reference's
// class (LumenPT/src/Tools/LumenPTModelConverter.h:10-25: LoadFile, ConvertGLTF, SetRendererRef, ms_ExtensionName), so that the adapter's
// OpenCustomFileFormat / CreateCustomFileFormat compile and run on a machine without the reference tree.  Inside the reference tree the adapter uses the
// reference's own converter instead (it only needs a LumenRenderer&).
//
// Behaviour follows LoadFile / LoadNode / SetRendererRef (LumenPTModelConverter.cpp:72-334): images are decoded to RGBA8, the G channel of a
// metal-roughness map is clamped to >= 1, base-colour and emissive maps are created with normalize = true (sRGB decode) and every other map without, absent
// maps fall back to four 1x1 defaults, vertices arrive interleaved as the file holds them (64-byte `Vertex` records: lumen_min.h) with V as stored (no flip), every node with a mesh becomes one mesh instance
// with the world matrix parent * local composed in float (glm operation order, Transform.cpp:282-308).  File layout: lumenrenderer_amd/ollad.py.
// Limits of this tree: PNG images only (8 / 16 bit, non-interlaced; inflated with zlib — the reference decodes with stb_image, which is not here), and no
// glTF -> .ollad conversion (no JSON / JPEG code here): ConvertGLTF returns an empty resource and the caller falls back as SceneManager::LoadGLTF does.
#pragma once
#include "../lumen_min.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

class LumenPTModelConverter
{
public:
    static inline const std::string ms_ExtensionName = ".ollad";

    void SetRendererRef(LumenRenderer& renderer)
    {
        m_RendererRef = &renderer;
        uint8_t white[4] = {255, 255, 255, 255}, normal[4] = {128, 128, 255, 0};
        m_DefaultWhiteTexture = renderer.CreateTexture(white, 1, 1, true);
        m_DefaultMetalRoughnessTexture = renderer.CreateTexture(white, 1, 1, false);
        m_DefaultNormalTexture = renderer.CreateTexture(normal, 1, 1, false);
        m_DefaultEmissiveTexture = renderer.CreateTexture(white, 1, 1, true);
    }

    Lumen::SceneManager::GLTFResource ConvertGLTF(std::string) { return {}; }

    Lumen::SceneManager::GLTFResource LoadFile(std::string path)
    {
        Lumen::SceneManager::GLTFResource res;
        std::ifstream f(path, std::ios::binary);
        if (!f || !m_RendererRef) return res;                                   // empty path = no file (the reference's contract)
        std::vector<char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        m_Data = file.data(); m_Size = file.size(); m_Pos = 0; m_Bad = false;
        const uint64_t headerSize = Take<uint64_t>();
        if (m_Bad || headerSize > m_Size - 8) return res;
        const char* blob = m_Data + 8 + headerSize;
        const uint64_t blobSize = m_Size - 8 - headerSize;

        std::vector<std::shared_ptr<Lumen::ILumenTexture>> textures;
        const uint64_t nTex = Take<uint64_t>();
        for (uint64_t i = 0; i < nTex && !m_Bad; i++) {
            const uint64_t off = Take<uint64_t>(), size = Take<uint64_t>(), type = Take<uint64_t>();
            if (off + size > blobSize) { m_Bad = true; break; }
            uint32_t w = 0, h = 0;
            std::vector<uint8_t> px;
            if (!DecodePng(reinterpret_cast<const uint8_t*>(blob + off), size, px, w, h)) { std::fprintf(stderr, "[ollad] image %llu is not a PNG this reader decodes\n", static_cast<unsigned long long>(i)); m_Bad = true; break; }
            if (type == 4) for (size_t k = 0; k < static_cast<size_t>(w) * h; k++) if (px[4 * k + 1] < 1) px[4 * k + 1] = 1;       // EMetalRoughness: roughness >= 1/255
            textures.push_back(m_RendererRef->CreateTexture(px.data(), w, h, type == 1 || type == 3));                                    // EDiffuse, EEmissive: sRGB
        }
        const uint64_t nMat = Take<uint64_t>();
        for (uint64_t i = 0; i < nMat && !m_Bad; i++) {
            struct { float color[4], emission[3]; int32_t tex[8]; float scalar[13], tint[3], transmittance[3]; } hm;
            static_assert(sizeof hm == 136, "HeaderMaterial");
            Bytes(&hm, sizeof hm);
            if (m_Bad) break;
            auto pick = [&](int32_t id, const std::shared_ptr<Lumen::ILumenTexture>& dflt) { return id != -1 && static_cast<size_t>(id) < textures.size() ? textures[static_cast<size_t>(id)] : dflt; };
            LumenRenderer::MaterialData d;
            d.m_DiffuseColor = glm::vec4(hm.color[0], hm.color[1], hm.color[2], hm.color[3]);
            d.m_EmissionVal = glm::vec3(hm.emission[0], hm.emission[1], hm.emission[2]);
            d.m_DiffuseTexture = pick(hm.tex[0], m_DefaultWhiteTexture); d.m_NormalMap = pick(hm.tex[1], m_DefaultNormalTexture);
            d.m_MetallicRoughnessTexture = pick(hm.tex[2], m_DefaultMetalRoughnessTexture); d.m_EmissiveTexture = pick(hm.tex[3], m_DefaultEmissiveTexture);
            d.m_TransmissionTexture = pick(hm.tex[4], m_DefaultWhiteTexture); d.m_ClearCoatTexture = pick(hm.tex[5], m_DefaultWhiteTexture);
            d.m_ClearCoatRoughnessTexture = pick(hm.tex[6], m_DefaultWhiteTexture); d.m_TintTexture = pick(hm.tex[7], m_DefaultWhiteTexture);
            d.m_TransmissionFactor = hm.scalar[0]; d.m_ClearCoatFactor = hm.scalar[1]; d.m_ClearCoatRoughnessFactor = hm.scalar[2]; d.m_IndexOfRefraction = hm.scalar[3];
            d.m_SpecularFactor = hm.scalar[4]; d.m_SpecularTintFactor = hm.scalar[5]; d.m_SubSurfaceFactor = hm.scalar[6]; d.m_Luminance = hm.scalar[7];
            d.m_Anisotropic = hm.scalar[8]; d.m_SheenFactor = hm.scalar[9]; d.m_SheenTintFactor = hm.scalar[10]; d.m_MetallicFactor = hm.scalar[11]; d.m_RoughnessFactor = hm.scalar[12];
            d.m_TintFactor = glm::vec3(hm.tint[0], hm.tint[1], hm.tint[2]);
            d.m_Transmittance = glm::vec3(hm.transmittance[0], hm.transmittance[1], hm.transmittance[2]);
            res.m_MaterialPool.push_back(m_RendererRef->CreateMaterial(d));
        }
        const uint64_t nMesh = Take<uint64_t>();
        for (uint64_t i = 0; i < nMesh && !m_Bad; i++) {
            const uint32_t nPrim = Take<uint32_t>();
            std::vector<std::shared_ptr<Lumen::ILumenPrimitive>> prims;
            for (uint32_t j = 0; j < nPrim && !m_Bad; j++) {
                const uint64_t vOff = Take<uint64_t>(), vSize = Take<uint64_t>(), iOff = Take<uint64_t>(), iSize = Take<uint64_t>();
                const uint32_t indexSize = Take<uint32_t>(), material = Take<uint32_t>();
                if (m_Bad || vOff + vSize > blobSize || iOff + iSize > blobSize || material >= res.m_MaterialPool.size() || (indexSize != 2 && indexSize != 4) || vSize % sizeof(Vertex) != 0) { m_Bad = true; break; }
                LumenRenderer::PrimitiveData d;
                d.m_Interleaved = true;
                d.m_IndexSize = indexSize;
                d.m_IndexBinary.assign(reinterpret_cast<const uint8_t*>(blob + iOff), reinterpret_cast<const uint8_t*>(blob + iOff + iSize));
                d.m_VertexBinary.assign(reinterpret_cast<const uint8_t*>(blob + vOff), reinterpret_cast<const uint8_t*>(blob + vOff + vSize));
                d.m_Material = res.m_MaterialPool[material];
                prims.push_back(m_RendererRef->CreatePrimitive(d));
            }
            if (!m_Bad) res.m_MeshPool.push_back(m_RendererRef->CreateMesh(prims));
        }
        const uint64_t nScene = Take<uint64_t>();
        for (uint64_t i = 0; i < nScene && !m_Bad; i++) {
            const uint32_t nRoots = Take<uint32_t>(), nameLength = Take<uint32_t>();
            res.m_Scenes.push_back(m_RendererRef->CreateScene());
            res.m_Scenes.back()->m_Name = Name(nameLength);
            glm::mat4 identity(1.0f);
            for (uint32_t j = 0; j < nRoots && !m_Bad; j++) LoadNode(res, *res.m_Scenes.back(), identity);
        }
        if (!m_Bad) res.m_Path = path;
        else res = Lumen::SceneManager::GLTFResource();
        return res;
    }

private:
    template <class T> T Take() { T v{}; Bytes(&v, sizeof v); return v; }
    void Bytes(void* dst, size_t n) { if (m_Bad || m_Pos + n > m_Size) { m_Bad = true; std::memset(dst, 0, n); return; } std::memcpy(dst, m_Data + m_Pos, n); m_Pos += n; }
    std::string Name(uint32_t n) { if (m_Bad || m_Pos + n > m_Size) { m_Bad = true; return {}; } std::string s(m_Data + m_Pos, n); m_Pos += n; return s; }

    void LoadNode(Lumen::SceneManager::GLTFResource& res, Lumen::ILumenScene& scene, const glm::mat4& parentWorld)
    {
        const uint32_t nameLength = Take<uint32_t>(), nChildren = Take<uint32_t>();
        float m[16]; Bytes(m, sizeof m);
        const int32_t meshId = Take<int32_t>();
        const std::string name = Name(nameLength);
        if (m_Bad) return;
        glm::mat4 local;                                                        // glm::make_mat4: 16 floats, column-major
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) local[c][r] = m[4 * c + r];
        glm::mat4 world;                                                        // parent * local, glm's operation order per element
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) {
            float acc = parentWorld[0][r] * local[c][0];
            acc = acc + parentWorld[1][r] * local[c][1];
            acc = acc + parentWorld[2][r] * local[c][2];
            acc = acc + parentWorld[3][r] * local[c][3];
            world[c][r] = acc;
        }
        if (meshId != -1) {
            if (static_cast<size_t>(meshId) >= res.m_MeshPool.size()) { m_Bad = true; return; }
            Lumen::MeshInstance* inst = scene.AddMesh();
            inst->m_Name = name;
            inst->SetMesh(res.m_MeshPool[static_cast<size_t>(meshId)]);
            inst->m_Transform = world;
            world = local;      // reference quirk (LoadNode :296-297): only the mesh INSTANCE is attached to the parent; the node's own transform, which its children hang on, is not
        }
        for (uint32_t i = 0; i < nChildren && !m_Bad; i++) LoadNode(res, scene, world);
    }

    // PNG: 8 / 16-bit grey, grey + alpha, RGB, RGBA or 8-bit palette (+ tRNS), non-interlaced -> RGBA8
    static uint32_t Be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
    static bool DecodePng(const uint8_t* p, uint64_t n, std::vector<uint8_t>& rgba, uint32_t& w, uint32_t& h)
    {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
        if (n < 8 || std::memcmp(p, sig, 8) != 0) return false;
        std::vector<uint8_t> idat, palette, trns;
        int depth = 0, type = 0, interlace = 0;
        for (uint64_t at = 8; at + 12 <= n;) {
            const uint32_t len = Be32(p + at);
            if (at + 12 + len > n) return false;
            const uint8_t* body = p + at + 8;
            if (!std::memcmp(p + at + 4, "IHDR", 4) && len >= 13) { w = Be32(body); h = Be32(body + 4); depth = body[8]; type = body[9]; interlace = body[12]; }
            else if (!std::memcmp(p + at + 4, "PLTE", 4)) palette.assign(body, body + len);
            else if (!std::memcmp(p + at + 4, "tRNS", 4)) trns.assign(body, body + len);
            else if (!std::memcmp(p + at + 4, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
            else if (!std::memcmp(p + at + 4, "IEND", 4)) break;
            at += 12 + len;
        }
        const int channels = type == 0 ? 1 : type == 2 ? 3 : type == 3 ? 1 : type == 4 ? 2 : type == 6 ? 4 : 0;
        if (!w || !h || !channels || interlace || (depth != 8 && depth != 16) || (type == 3 && depth != 8)) return false;
        const size_t bpp = static_cast<size_t>(channels) * (depth / 8), stride = bpp * w;
        std::vector<uint8_t> raw((stride + 1) * h);
        uLongf rawSize = static_cast<uLongf>(raw.size());
        if (uncompress(raw.data(), &rawSize, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || rawSize != raw.size()) return false;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        rgba.assign(static_cast<size_t>(w) * h * 4, 255);
        for (uint32_t y = 0; y < h; y++) {
            const uint8_t* row = raw.data() + (stride + 1) * y;
            const int filter = row[0];
            for (size_t x = 0; x < stride; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int pred = 0;
                if (filter == 1) pred = a; else if (filter == 2) pred = b; else if (filter == 3) pred = (a + b) / 2;
                else if (filter == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
                else if (filter != 0) return false;
                cur[x] = static_cast<uint8_t>(row[1 + x] + pred);
            }
            for (uint32_t x = 0; x < w; x++) {
                uint8_t* o = &rgba[(static_cast<size_t>(y) * w + x) * 4];
                const uint8_t* s = &cur[x * bpp];
                const int step = depth / 8;                                     // 16-bit samples: the high byte
                if (type == 0) { o[0] = o[1] = o[2] = s[0]; }
                else if (type == 2) { o[0] = s[0]; o[1] = s[step]; o[2] = s[2 * step]; }
                else if (type == 3) { const size_t k = s[0]; if (3 * k + 2 >= palette.size()) return false; o[0] = palette[3 * k]; o[1] = palette[3 * k + 1]; o[2] = palette[3 * k + 2]; o[3] = k < trns.size() ? trns[k] : 255; }
                else if (type == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[step]; }
                else { o[0] = s[0]; o[1] = s[step]; o[2] = s[2 * step]; o[3] = s[3 * step]; }
            }
            prev.swap(cur);
        }
        return true;
    }

    LumenRenderer* m_RendererRef = nullptr;
    std::shared_ptr<Lumen::ILumenTexture> m_DefaultWhiteTexture, m_DefaultMetalRoughnessTexture, m_DefaultNormalTexture, m_DefaultEmissiveTexture;
    const char* m_Data = nullptr;
    size_t m_Size = 0, m_Pos = 0;
    bool m_Bad = false;
};
