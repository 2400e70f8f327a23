// adapter_record.cpp — build-container test aid (tests/test_cpu_host.py): the reference's own SceneManager::LoadGLTF (Lumen/src/Lumen/ModelLoading/SceneManager.cpp:42-75)
// driving include/lumen_mi_renderer.hpp.  LoadGLTF asks the renderer first — OpenCustomFileFormat, then CreateCustomFileFormat — and the adapter answers with the
// reference's own LumenPTModelConverter (glTF -> .ollad -> LoadFile), whose every CreateTexture / CreateMaterial / CreatePrimitive / CreateMesh / CreateScene call lands
// in the adapter's virtuals and from there in the C ABI (resource creation is host-side: no GPU needed).  This subclass logs each call before forwarding it, then the
// instances of the loaded scene; the test compares the log with what lumenrenderer_amd/ollad.py reads from the .ollad file the run wrote.
//     adapter_record <directory/> <file.gltf>
#include "lumen_mi_renderer.hpp"
#include "Lumen/ModelLoading/SceneManager.h"

#include <cinttypes>
#include <cstddef>
#include <cstdio>
#include <cstring>

static uint64_t Fnv(const void* p, size_t n) { uint64_t h = 1469598103934665603ull; const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
static uint32_t Bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

class Recorder : public MI355X::Renderer
{
public:
    std::shared_ptr<Lumen::ILumenTexture> CreateTexture(void* px, uint32_t w, uint32_t h, bool normalize) override
    {
        uint64_t sum = 0;
        for (size_t i = 0; i < static_cast<size_t>(w) * h * 4; i++) sum += static_cast<const uint8_t*>(px)[i];
        std::printf("tex %u %u %d %016" PRIx64 " %" PRIu64 "\n", w, h, normalize ? 1 : 0, Fnv(px, static_cast<size_t>(w) * h * 4), sum);
        return MI355X::Renderer::CreateTexture(px, w, h, normalize);
    }
    std::shared_ptr<Lumen::ILumenMaterial> CreateMaterial(const MaterialData& d) override
    {
        const float f[] = {d.m_DiffuseColor.x, d.m_DiffuseColor.y, d.m_DiffuseColor.z, d.m_DiffuseColor.w, d.m_EmissionVal.x, d.m_EmissionVal.y, d.m_EmissionVal.z,
                           d.m_TransmissionFactor, d.m_ClearCoatFactor, d.m_ClearCoatRoughnessFactor, d.m_IndexOfRefraction, d.m_SpecularFactor, d.m_SpecularTintFactor,
                           d.m_SubSurfaceFactor, d.m_Luminance, d.m_Anisotropic, d.m_SheenFactor, d.m_SheenTintFactor, d.m_MetallicFactor, d.m_RoughnessFactor,
                           d.m_TintFactor.x, d.m_TintFactor.y, d.m_TintFactor.z, d.m_Transmittance.x, d.m_Transmittance.y, d.m_Transmittance.z};
        std::printf("mat");
        for (float v : f) std::printf(" %08x", Bits(v));
        std::printf("\n");
        return MI355X::Renderer::CreateMaterial(d);
    }
    std::unique_ptr<Lumen::ILumenPrimitive> CreatePrimitive(PrimitiveData& d) override
    {
        std::vector<uint32_t> idx(d.m_IndexBinary.size() / d.m_IndexSize);
        for (size_t i = 0; i < idx.size(); i++) idx[i] = d.m_IndexSize == 2 ? reinterpret_cast<const uint16_t*>(d.m_IndexBinary.data())[i] : reinterpret_cast<const uint32_t*>(d.m_IndexBinary.data())[i];
        // the 12 floats of every vertex (position uv normal tangent), whatever padding sizeof(Vertex) carries
        std::vector<float> v12;
        for (size_t i = 0; i + sizeof(Vertex) <= d.m_VertexBinary.size(); i += sizeof(Vertex)) {
            const uint8_t* src = d.m_VertexBinary.data() + i;
            float f[12];
            std::memcpy(f, src + offsetof(Vertex, m_Position), 12); std::memcpy(f + 3, src + offsetof(Vertex, m_UVCoord), 8);
            std::memcpy(f + 5, src + offsetof(Vertex, m_Normal), 12); std::memcpy(f + 8, src + offsetof(Vertex, m_Tangent), 16);
            v12.insert(v12.end(), f, f + 12);
        }
        std::printf("prim %d %zu %zu %zu %zu %016" PRIx64 " %016" PRIx64 "\n", d.m_Interleaved ? 1 : 0, sizeof(Vertex), v12.size() / 12, idx.size(), d.m_IndexSize,
                    Fnv(v12.data(), v12.size() * 4), Fnv(idx.data(), idx.size() * 4));
        auto p = MI355X::Renderer::CreatePrimitive(d);
        std::printf("lights %u\n", p->m_NumLights);
        return p;
    }
    std::shared_ptr<Lumen::ILumenMesh> CreateMesh(std::vector<std::shared_ptr<Lumen::ILumenPrimitive>>& prims) override
    {
        std::printf("mesh %zu\n", prims.size());
        return MI355X::Renderer::CreateMesh(prims);
    }
};

int main(int argc, char** argv)
{
    if (argc != 3) return 64;
    Recorder renderer;                                   // no Init: this machine has no GPU; model loading does not need one
    Lumen::SceneManager manager;
    manager.SetPipeline(renderer);
    auto* res = manager.LoadGLTF(argv[2], argv[1]);
    if (!res || res->m_Path.empty() || res->m_Scenes.empty()) { std::fprintf(stderr, "LoadGLTF returned nothing\n"); return 65; }
    std::printf("path %s\n", res->m_Path.c_str());
    for (auto& inst : res->m_Scenes[0]->m_MeshInstances) {
        const glm::mat4 w = inst->m_Transform.GetWorldTransformationMatrix();
        std::printf("inst");
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) std::printf(" %08x", Bits(w[c][r]));      // row-major
        std::printf("\n");
    }
    return 0;
}
