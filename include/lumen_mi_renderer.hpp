// lumen_mi_renderer.hpp — header-only C++ adapter: the reference's own class shape on top of the C ABI.
//
// Drop this file (and lumen_mi.h) into the reference tree, link liblumen_mi.so, and construct
// `MI355X::Renderer` where Sandbox constructs `LumenPT` (LumenPT/src/LumenPT.h:12, Sandbox/src/Application.cpp:83):
//
//     auto renderer = std::make_shared<MI355X::Renderer>();
//     MI355X::Renderer::Settings s; s.depth = 5; s.renderResolution = {1280, 720}; ...
//     renderer->Init(s);                       // WaveFrontRenderer::Init(const WaveFrontSettings&)  (Application.cpp:84-95)
//     renderer->CreateDefaultResources(); ...  // everything after this line is the unchanged Sandbox code
//
// It derives from the reference's LumenRenderer (Lumen/src/Lumen/Renderer/LumenRenderer.h:37-219) and returns
// objects implementing ILumenTexture / ILumenMaterial / ILumenPrimitive / ILumenMesh / ILumenScene
// (ILumenResources.h:12-127, ILumenScene.h:11-71), so SceneManager, OutputLayer and the tool UI keep working.
// This header includes only reference headers and lumen_mi.h — no HIP.  It cannot be compiled outside the
// reference tree (it needs Lumen's headers and glm); INTEGRATION.md describes the build hook.
#pragma once
#include "lumen_mi.h"

#include "Lumen/Renderer/LumenRenderer.h"
#include "Lumen/Renderer/ILumenResources.h"
#include "Lumen/ModelLoading/ILumenScene.h"
#include "Lumen/ModelLoading/MeshInstance.h"
#include "Lumen/Renderer/Camera.h"
#include "Tools/FrameSnapshot.h"            // LumenPT/src: EndSnapshot() returns a unique_ptr of the complete type
#include "Shaders/CppCommon/ModelStructs.h" // LumenPT/src: struct Vertex — what an interleaved PrimitiveData holds (64 bytes under CUDA's vector-type alignment)
#include "Tools/LumenPTModelConverter.h"    // LumenPT/src: the reference's own .ollad reader / glTF converter — plain host C++ that only needs a LumenRenderer&

#include <glm/glm.hpp>
#include <glm/gtc/type_ptr.hpp>

#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace MI355X
{
    inline void Check(int rc, const char* what)
    {
        // the reference has no error returns: it asserts / aborts (CudaUtilities.h:24-28); keep that contract at this level
        if (rc != LUMEN_MI_OK && rc != LUMEN_MI_NO_LIGHTS) { std::fprintf(stderr, "[lumen_mi] %s failed (%d): %s\n", what, rc, lumen_mi_last_error()); std::abort(); }
    }

    class Texture : public Lumen::ILumenTexture
    {
    public:
        explicit Texture(lumen_mi_handle h) : m_Handle(h) {}
        lumen_mi_handle m_Handle;
    };

    // ILumenMaterial: 25 setters + 19 getters; every setter re-sends the whole MaterialData (lumen_mi_update_material)
    class Material : public Lumen::ILumenMaterial
    {
    public:
        Material(lumen_mi_renderer* r, const LumenRenderer::MaterialData& d) : m_R(r), m_Data(d) { auto c = ToC(); Check(lumen_mi_create_material(m_R, &c, &m_Handle), "create_material"); }

        void SetDiffuseColor(const glm::vec4& v) override { m_Data.m_DiffuseColor = v; Push(); }
        void SetDiffuseTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_DiffuseTexture = t; Push(); }
        void SetEmission(const glm::vec3& v = glm::vec3(0.f)) override { m_Data.m_EmissionVal = v; Push(); }
        void SetEmissiveTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_EmissiveTexture = t; Push(); }
        void SetMetalRoughnessTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_MetallicRoughnessTexture = t; Push(); }
        void SetNormalTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_NormalMap = t; Push(); }
        void SetClearCoatTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_ClearCoatTexture = t; Push(); }
        void SetClearCoatRoughnessTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_ClearCoatRoughnessTexture = t; Push(); }
        void SetClearCoatFactor(float f) override { m_Data.m_ClearCoatFactor = f; Push(); }
        void SetClearCoatRoughnessFactor(float f) override { m_Data.m_ClearCoatRoughnessFactor = f; Push(); }
        void SetLuminance(float f) override { m_Data.m_Luminance = f; Push(); }
        void SetSheenFactor(float f) override { m_Data.m_SheenFactor = f; Push(); }
        void SetSheenTintFactor(float f) override { m_Data.m_SheenTintFactor = f; Push(); }
        void SetAnisotropic(float f) override { m_Data.m_Anisotropic = f; Push(); }
        void SetTintTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_TintTexture = t; Push(); }
        void SetTintFactor(const glm::vec3& v) override { m_Data.m_TintFactor = v; Push(); }
        void SetTransmissionTexture(std::shared_ptr<Lumen::ILumenTexture> t) override { m_Data.m_TransmissionTexture = t; Push(); }
        void SetTransmissionFactor(float f) override { m_Data.m_TransmissionFactor = f; Push(); }
        void SetTransmittanceFactor(const glm::vec3& v) override { m_Data.m_Transmittance = v; Push(); }
        void SetIndexOfRefraction(float f) override { m_Data.m_IndexOfRefraction = f; Push(); }
        void SetSpecularFactor(float f) override { m_Data.m_SpecularFactor = f; Push(); }
        void SetSpecularTintFactor(float f) override { m_Data.m_SpecularTintFactor = f; Push(); }
        void SetSubSurfaceFactor(float f) override { m_Data.m_SubSurfaceFactor = f; Push(); }
        void SetMetallicFactor(float f) override { m_Data.m_MetallicFactor = f; Push(); }
        void SetRoughnessFactor(float f) override { m_Data.m_RoughnessFactor = f; Push(); }

        float GetClearCoatFactor() override { return m_Data.m_ClearCoatFactor; }
        float GetClearCoatRoughnessFactor() override { return m_Data.m_ClearCoatRoughnessFactor; }
        float GetLuminance() override { return m_Data.m_Luminance; }
        float GetSheenFactor() override { return m_Data.m_SheenFactor; }
        float GetSheenTintFactor() override { return m_Data.m_SheenTintFactor; }
        float GetAnisotropic() override { return m_Data.m_Anisotropic; }
        glm::vec3 GetTintFactor() override { return m_Data.m_TintFactor; }
        float GetTransmissionFactor() override { return m_Data.m_TransmissionFactor; }
        glm::vec3 GetTransmittanceFactor() override { return m_Data.m_Transmittance; }
        float GetIndexOfRefraction() override { return m_Data.m_IndexOfRefraction; }
        float GetSpecularFactor() override { return m_Data.m_SpecularFactor; }
        float GetSpecularTintFactor() override { return m_Data.m_SpecularTintFactor; }
        float GetSubSurfaceFactor() override { return m_Data.m_SubSurfaceFactor; }
        float GetMetallicFactor() override { return m_Data.m_MetallicFactor; }
        float GetRoughnessFactor() override { return m_Data.m_RoughnessFactor; }
        glm::vec4 GetDiffuseColor() const override { return m_Data.m_DiffuseColor; }
        glm::vec3 GetEmissiveColor() const override { return m_Data.m_EmissionVal; }
        Lumen::ILumenTexture& GetDiffuseTexture() const override { return *m_Data.m_DiffuseTexture; }
        Lumen::ILumenTexture& GetEmissiveTexture() const override { return *m_Data.m_EmissiveTexture; }

        lumen_mi_handle m_Handle = 0;

    private:
        static lumen_mi_handle H(const std::shared_ptr<Lumen::ILumenTexture>& t) { return t ? static_cast<Texture*>(t.get())->m_Handle : 0; }
        lumen_mi_material_data ToC() const
        {
            lumen_mi_material_data c{};
            for (int i = 0; i < 4; i++) c.diffuse_color[i] = m_Data.m_DiffuseColor[i];
            for (int i = 0; i < 3; i++) { c.emission[i] = m_Data.m_EmissionVal[i]; c.tint_factor[i] = m_Data.m_TintFactor[i]; c.transmittance[i] = m_Data.m_Transmittance[i]; }
            c.diffuse_texture = H(m_Data.m_DiffuseTexture); c.normal_map = H(m_Data.m_NormalMap);
            c.metallic_roughness_texture = H(m_Data.m_MetallicRoughnessTexture); c.emissive_texture = H(m_Data.m_EmissiveTexture);
            c.transmission_texture = H(m_Data.m_TransmissionTexture); c.clearcoat_texture = H(m_Data.m_ClearCoatTexture);
            c.clearcoat_roughness_texture = H(m_Data.m_ClearCoatRoughnessTexture); c.tint_texture = H(m_Data.m_TintTexture);
            c.transmission_factor = m_Data.m_TransmissionFactor; c.clearcoat_factor = m_Data.m_ClearCoatFactor;
            c.clearcoat_roughness_factor = m_Data.m_ClearCoatRoughnessFactor; c.index_of_refraction = m_Data.m_IndexOfRefraction;
            c.specular_factor = m_Data.m_SpecularFactor; c.specular_tint_factor = m_Data.m_SpecularTintFactor; c.subsurface_factor = m_Data.m_SubSurfaceFactor;
            c.luminance = m_Data.m_Luminance; c.anisotropic = m_Data.m_Anisotropic; c.sheen_factor = m_Data.m_SheenFactor; c.sheen_tint_factor = m_Data.m_SheenTintFactor;
            c.metallic_factor = m_Data.m_MetallicFactor; c.roughness_factor = m_Data.m_RoughnessFactor;
            return c;
        }
        void Push() { auto c = ToC(); Check(lumen_mi_update_material(m_R, m_Handle, &c), "update_material"); }
        lumen_mi_renderer* m_R;
        LumenRenderer::MaterialData m_Data;
    };

    class Primitive : public Lumen::ILumenPrimitive { public: lumen_mi_handle m_Handle = 0; };

    class Mesh : public Lumen::ILumenMesh
    {
    public:
        Mesh(std::vector<std::shared_ptr<Lumen::ILumenPrimitive>>& prims, lumen_mi_handle h) : ILumenMesh(prims), m_Handle(h) {}
        lumen_mi_handle m_Handle;
    };

    class MeshInstance : public Lumen::MeshInstance
    {
    public:
        MeshInstance(lumen_mi_renderer* r, lumen_mi_handle scene) : m_R(r), m_Scene(scene) {}
        void SetMesh(std::shared_ptr<Lumen::ILumenMesh> mesh) override
        {
            Lumen::MeshInstance::SetMesh(mesh);
            Check(lumen_mi_scene_add_mesh(m_R, m_Scene, static_cast<Mesh*>(mesh.get())->m_Handle, &m_Handle), "scene_add_mesh");
            PushOverride();                    // an override material set before the mesh (the instance had no native handle yet)
        }
        void SetEmissiveness(const Emissiveness& e) override { Lumen::MeshInstance::SetEmissiveness(e); PushEmissiveness(); }
        void SetOverrideMaterial(std::shared_ptr<Lumen::ILumenMaterial> m) override
        {
            Lumen::MeshInstance::SetOverrideMaterial(m);
            PushOverride();
        }
        // called by Renderer before every frame: the app edits m_Transform directly (the reference polls dirty flags, PTMeshInstance.cpp:123-178)
        void Sync()
        {
            if (!m_Handle) return;
            const glm::mat4 rowMajor = glm::transpose(m_Transform.GetWorldTransformationMatrix());     // PTMeshInstance.cpp:147-151
            Check(lumen_mi_instance_set_transform(m_R, m_Handle, glm::value_ptr(rowMajor)), "set_transform");
            PushEmissiveness();
            PushOverride();
        }
        lumen_mi_handle m_Handle = 0;

    private:
        void PushEmissiveness()
        {
            if (!m_Handle) return;
            const float rad[3] = {m_EmissiveProperties.m_OverrideRadiance.x, m_EmissiveProperties.m_OverrideRadiance.y, m_EmissiveProperties.m_OverrideRadiance.z};
            Check(lumen_mi_instance_set_emissiveness(m_R, m_Handle, static_cast<int>(m_EmissiveProperties.m_EmissionMode), rad, m_EmissiveProperties.m_Scale), "set_emissiveness");
        }
        void PushOverride()
        {
            // a null override means "use the mesh's own materials" (MeshInstance.h:57-65, PTMeshInstance.cpp:163-165): pushed as handle 0,
            // which clears an override set earlier
            if (!m_Handle) return;
            Check(lumen_mi_instance_set_override_material(m_R, m_Handle, m_OverrideMaterial ? static_cast<Material*>(m_OverrideMaterial.get())->m_Handle : 0), "set_override_material");
        }
        lumen_mi_renderer* m_R;
        lumen_mi_handle m_Scene;
    };

    class Scene : public Lumen::ILumenScene
    {
    public:
        Scene(lumen_mi_renderer* r, const LumenRenderer::SceneData& d) : ILumenScene(d.m_CameraPosition, d.m_CameraUp), m_R(r) { Check(lumen_mi_create_scene(m_R, &m_Handle), "create_scene"); }
        Lumen::MeshInstance* AddMesh() override
        {
            m_MeshInstances.push_back(std::make_unique<MeshInstance>(m_R, m_Handle));
            return m_MeshInstances.back().get();
        }
        void Clear() override { ILumenScene::Clear(); Check(lumen_mi_scene_clear(m_R, m_Handle), "scene_clear"); }
        lumen_mi_handle m_Handle = 0;

    private:
        lumen_mi_renderer* m_R;
    };

    class Renderer : public LumenRenderer
    {
    public:
        // WaveFrontSettings (LumenPT/src/Framework/WaveFrontRenderer.h:31-48) without the PTX paths
        // renderThread: true (the reference's behaviour) = StartRendering starts a render thread that loops TraceFrame (WaveFrontRenderer.cpp:1109-1117);
        // false = frames are traced from PerformDeferredOperations on the caller's thread, one per call (deterministic frame counts: tests, offline rendering)
        struct Settings { unsigned depth = 5; glm::uvec2 renderResolution{1280, 720}; glm::uvec2 outputResolution{1280, 720}; bool blendOutput = false; int device = 0; bool renderThread = true; };

        Renderer() { Check(lumen_mi_create(&m_R), "create"); }
        ~Renderer() override { if (m_ThreadRunning) lumen_mi_stop_rendering(m_R); if (m_Group) lumen_mi_group_destroy(m_Group); lumen_mi_destroy(m_R); }      // WaveFrontRenderer.cpp:1360-1371: the thread is joined first

        // Tile group (new functionality: the reference is single-GPU): this process becomes rank `rank` of `world` processes, one per GPU, that render ONE image.  Call after
        // Init (at the full image resolution) on every rank; `id` = the LUMEN_MI_GROUP_ID_BYTES rank 0 obtained from GroupUniqueId(), handed over by the application (file,
        // environment, socket).  From then on TraceFrame renders this rank's tile + halo, exchanges the seam history where the path depth needs it and gathers the tiles on
        // rank 0 over RCCL (csrc/group.cpp), and rank 0's GetOutputTexturePixels returns the stitched frame; other ranks return no pixels.  Frames are traced from
        // PerformDeferredOperations on the caller's thread (every rank must trace the same number of frames: a free-running render thread per rank could not promise that).
        static std::vector<uint8_t> GroupUniqueId() { std::vector<uint8_t> id(LUMEN_MI_GROUP_ID_BYTES); Check(lumen_mi_group_unique_id(id.data()), "group_unique_id"); return id; }
        void SetGroup(uint32_t rank, uint32_t world, const std::vector<uint8_t>& id, const lumen_mi_transport* hostTransport = nullptr)
        {
            if (m_ThreadRunning) throw std::runtime_error("MI355X::Renderer::SetGroup: call before StartRendering");
            if (!id.empty() && id.size() != LUMEN_MI_GROUP_ID_BYTES) throw std::runtime_error("MI355X::Renderer::SetGroup: the id must be LUMEN_MI_GROUP_ID_BYTES long");
            if (m_Group) { Check(lumen_mi_group_destroy(m_Group), "group_destroy"); m_Group = nullptr; }
            Check(lumen_mi_group_create(m_R, rank, world, id.empty() ? nullptr : id.data(), hostTransport, &m_Group), "group_create");
            float ms = 0.f;
            Check(lumen_mi_group_self_test(m_Group, &ms), "group_self_test");
            m_GroupRank = rank; m_UseThread = false;
        }
        lumen_mi_group* NativeGroup() { return m_Group; }

        void Init(const Settings& s)
        {
            lumen_mi_settings c{s.depth, s.renderResolution.x, s.renderResolution.y, s.outputResolution.x, s.outputResolution.y, s.blendOutput ? 1 : 0, s.device};
            Check(lumen_mi_init(m_R, &c), "init");
            m_UseThread = s.renderThread;
            AttachModelConverter();                                         // WaveFrontRenderer.cpp:305
        }

        // The model cache the reference's SceneManager::LoadGLTF asks for FIRST (SceneManager.cpp:56-64): `<model>.ollad` beside the glTF, read — or written from
        // the glTF and then read — by LumenPTModelConverter, which creates every texture / material / primitive / mesh / scene through this renderer's own
        // virtuals: base-colour and emissive maps sRGB-decoded, no V flip, 48-byte interleaved vertices.  WaveFrontRenderer.cpp:1135-1146.
        Lumen::SceneManager::GLTFResource OpenCustomFileFormat(const std::string& originalFilePath) override
        {
            std::string p = originalFilePath;
            const size_t dot = p.find_last_of('.'), slash = p.find_last_of("/\\");
            if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) p.erase(dot);      // std::filesystem::path::replace_extension
            AttachModelConverter();
            return m_ModelConverter.LoadFile(p + LumenPTModelConverter::ms_ExtensionName);
        }
        Lumen::SceneManager::GLTFResource CreateCustomFileFormat(const std::string& originalFilePath) override { AttachModelConverter(); return m_ModelConverter.ConvertGLTF(originalFilePath); }

        // StartRendering (LumenRenderer.h:151): the render thread of the C ABI loops TraceFrame; the application keeps editing m_Scene on its own thread
        // and PerformDeferredOperations — called once per displayed frame by LumenApp::Run — pushes those edits (instance transforms, emissiveness, override
        // materials, camera) through the C ABI's setters, which serialise with the frame in flight; the next TraceFrame picks them up through the scene's
        // dirty flags, as the reference's does (PTMeshInstance.cpp:28-49,123-178; PTScene.cpp:62-72).
        void StartRendering() override
        {
            m_Started = true;
            if (!m_UseThread) return;
            PushSceneState();
            Check(lumen_mi_start_rendering(m_R), "start_rendering");
            m_ThreadRunning = true;
        }
        void PerformDeferredOperations() override
        {
            if (!m_Started) return;
            if (m_ThreadRunning) { PushSceneState(); CollectFrameStats(); }
            else TraceFrame();
        }

        std::unique_ptr<Lumen::ILumenPrimitive> CreatePrimitive(PrimitiveData& d) override
        {
            lumen_mi_primitive_data c{};
            // interleaved vertices are the reference's own `Vertex` records, whatever the compiler makes of them: 64 bytes with CUDA's aligned vector types
            // (the reference's build), 48 if a build ever packs it
            static_assert(sizeof(Vertex) == 64 || sizeof(Vertex) == 48, "unexpected Vertex layout");
            c.interleaved = d.m_Interleaved ? (sizeof(Vertex) == 64 ? LUMEN_MI_VERTICES_REFERENCE64 : LUMEN_MI_VERTICES_PACKED48) : LUMEN_MI_VERTICES_SEPARATE;
            c.vertex_binary = d.m_VertexBinary.data();
            c.positions = d.m_Positions.Empty() ? nullptr : reinterpret_cast<const float*>(&d.m_Positions[0]);      // (an interleaved primitive has no attribute views: VectorView::operator[] asserts on them)
            c.tex_coords = d.m_TexCoords.Empty() ? nullptr : reinterpret_cast<const float*>(&d.m_TexCoords[0]);
            c.normals = d.m_Normals.Empty() ? nullptr : reinterpret_cast<const float*>(&d.m_Normals[0]);
            c.tangents = d.m_Tangents.Empty() ? nullptr : reinterpret_cast<const float*>(&d.m_Tangents[0]);
            c.n_vertices = static_cast<uint32_t>(d.m_Interleaved ? d.m_VertexBinary.size() / sizeof(Vertex) : d.m_Positions.Size());
            c.index_binary = d.m_IndexBinary.data();
            c.index_size = static_cast<uint32_t>(d.m_IndexSize);
            c.n_indices = static_cast<uint32_t>(d.m_IndexBinary.size() / d.m_IndexSize);
            c.material = static_cast<Material*>(d.m_Material.get())->m_Handle;
            auto p = std::make_unique<Primitive>();
            uint32_t numLights = 0;
            Check(lumen_mi_create_primitive(m_R, &c, &p->m_Handle, &numLights), "create_primitive");
            p->m_Material = d.m_Material; p->m_NumLights = numLights; p->m_ContainEmissive = numLights > 0;
            return p;
        }
        std::shared_ptr<Lumen::ILumenMesh> CreateMesh(std::vector<std::shared_ptr<Lumen::ILumenPrimitive>>& prims) override
        {
            std::vector<lumen_mi_handle> hs;
            for (auto& p : prims) hs.push_back(static_cast<Primitive*>(p.get())->m_Handle);
            lumen_mi_handle h = 0;
            Check(lumen_mi_create_mesh(m_R, hs.data(), static_cast<uint32_t>(hs.size()), &h), "create_mesh");
            return std::make_shared<Mesh>(prims, h);
        }
        std::shared_ptr<Lumen::ILumenTexture> CreateTexture(void* px, uint32_t w, uint32_t h, bool normalize) override
        {
            lumen_mi_handle t = 0;
            Check(lumen_mi_create_texture(m_R, px, w, h, normalize ? 1 : 0, &t), "create_texture");
            return std::make_shared<Texture>(t);
        }
        std::shared_ptr<Lumen::ILumenMaterial> CreateMaterial(const MaterialData& d) override { return std::make_shared<Material>(m_R, d); }
        std::shared_ptr<Lumen::ILumenScene> CreateScene(SceneData d = {}) override { return std::make_shared<Scene>(m_R, d); }
        std::shared_ptr<Lumen::ILumenVolume> CreateVolume(const std::string&) override { return nullptr; }   // volumes are out of scope
        void InitNGX() override {}                                                                             // DLSS: out of scope

        unsigned int GetOutputTexture() override { return 0; }   // no GL interop: OutputLayer uploads GetOutputTexturePixels() instead
        std::vector<uint8_t> GetOutputTexturePixels(uint32_t& w, uint32_t& h) override
        {
            // size of the LAST TRACED frame, not the pending resolution (SetRenderResolution applies at the next frame): a first call
            // with capacity 0 reports it
            uint8_t none = 0;
            w = h = 0;
            if (m_Group) return GroupPixels(w, h);
            (void)lumen_mi_get_output_pixels(m_R, &none, 0, &w, &h);
            std::vector<uint8_t> px(static_cast<size_t>(w) * h * 4);
            if (!px.empty()) Check(lumen_mi_get_output_pixels(m_R, px.data(), px.size(), &w, &h), "get_output_pixels");
            return px;
        }
        void SetRenderResolution(glm::uvec2 r) override { Check(lumen_mi_set_render_resolution(m_R, r.x, r.y), "set_render_resolution"); }
        void SetOutputResolution(glm::uvec2 r) override { Check(lumen_mi_set_output_resolution(m_R, r.x, r.y), "set_output_resolution"); }
        glm::uvec2 GetRenderResolution() override { uint32_t w, h; lumen_mi_get_render_resolution(m_R, &w, &h); return {w, h}; }
        glm::uvec2 GetOutputResolution() override { uint32_t w, h; lumen_mi_get_output_resolution(m_R, &w, &h); return {w, h}; }
        void SetBlendMode(bool b) override { Check(lumen_mi_set_blend_mode(m_R, b ? 1 : 0), "set_blend_mode"); }
        bool GetBlendMode() const override { int b = 0; lumen_mi_get_blend_mode(m_R, &b); return b != 0; }
        void BeginSnapshot() override {}
        std::unique_ptr<FrameSnapshot> EndSnapshot() override { return nullptr; }

        // body of WaveFrontRenderer::TraceFrame: push the app-side scene state, then render one frame
        void TraceFrame()
        {
            if (!PushSceneState()) return;
            if (m_Group) { Check(lumen_mi_group_trace_frame(m_Group), "group_trace_frame"); Check(lumen_mi_group_gather(m_Group), "group_gather"); m_GroupFrames++; return; }
            Check(lumen_mi_trace_frame(m_R), "trace_frame");
            CollectFrameStats();
        }

        lumen_mi_renderer* Native() { return m_R; }

    private:
        // what the application edits directly on m_Scene (instance transforms, camera), sent to the native scene
        bool PushSceneState()
        {
            if (!m_Scene) return false;
            auto* scene = static_cast<Scene*>(m_Scene.get());
            Check(lumen_mi_set_scene(m_R, scene->m_Handle), "set_scene");
            for (auto& mi : scene->m_MeshInstances) static_cast<MeshInstance*>(mi.get())->Sync();
            glm::mat4 prev, cur;
            scene->m_Camera->GetMatrixData(prev, cur);            // columns: right, up, forward, position (Camera.cpp:122-140)
            const float pos[3] = {cur[3].x, cur[3].y, cur[3].z}, right[3] = {cur[0].x, cur[0].y, cur[0].z}, up[3] = {cur[1].x, cur[1].y, cur[1].z}, fwd[3] = {cur[2].x, cur[2].y, cur[2].z};
            Check(lumen_mi_camera_set(m_R, pos, right, up, fwd, 90.0f), "camera_set");   // Camera::m_FovY is fixed at 90 (Camera.h:63)
            return true;
        }
        void CollectFrameStats()
        {
            uint64_t frames = 0;
            const bool haveCount = lumen_mi_get_frame_stat(m_R, "Frames Traced", &frames) == LUMEN_MI_OK;
            std::lock_guard<std::mutex> lk(m_FrameStatsMutex);
            static const char* keys[] = {"Wavefront Iteration", "Shadow Rays", "ReSTIR", "Total Frame Time"};
            for (const char* k : keys) { uint64_t us = 0; if (lumen_mi_get_frame_stat(m_R, k, &us) == LUMEN_MI_OK) m_LastFrameStats.m_Times[k] = us; }
            if (haveCount) m_LastFrameStats.m_Id = frames; else ++m_LastFrameStats.m_Id;      // FrameStats::m_Id counts traced frames (WaveFrontRenderer.cpp:1078-1083)
        }

        // the converter creates its four default 1x1 textures through this renderer (LumenPTModelConverter::SetRendererRef); resources are host-side objects of the
        // C ABI, so a model may be loaded before Init (once: the reference attaches in Init, WaveFrontRenderer.cpp:305)
        void AttachModelConverter() { if (!m_ConverterAttached) { m_ConverterAttached = true; m_ModelConverter.SetRendererRef(*this); } }

        // rank 0 of a tile group: the stitched fp32 frame through the sRGB transfer function (IEC 61966-2-1, 256 levels, clamped), alpha 255; other ranks: nothing
        std::vector<uint8_t> GroupPixels(uint32_t& w, uint32_t& h)
        {
            if (m_GroupRank != 0 || !m_GroupFrames) { Check(lumen_mi_group_synchronize(m_Group), "group_synchronize"); return {}; }
            lumen_mi_get_render_resolution(m_R, &w, &h);
            std::vector<float> frame(static_cast<size_t>(w) * h * 4);
            Check(lumen_mi_group_get_frame(m_Group, frame.data(), frame.size() * sizeof(float)), "group_get_frame");
            std::vector<uint8_t> px(frame.size());
            auto level = [](float c) { const float s = c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f; const float q = (s < 0.f ? 0.f : s > 1.f ? 1.f : s) * 256.f; return static_cast<uint8_t>(q > 255.f ? 255.f : q); };
            for (size_t i = 0; i < px.size(); i += 4) { px[i] = level(frame[i]); px[i + 1] = level(frame[i + 1]); px[i + 2] = level(frame[i + 2]); px[i + 3] = 255; }
            return px;
        }

        lumen_mi_renderer* m_R = nullptr;
        lumen_mi_group* m_Group = nullptr;
        uint32_t m_GroupRank = 0; uint64_t m_GroupFrames = 0;
        LumenPTModelConverter m_ModelConverter;
        bool m_UseThread = true, m_Started = false, m_ThreadRunning = false, m_ConverterAttached = false;
    };
}  // namespace MI355X
