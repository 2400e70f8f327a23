// lumen_min.h — the part of the Lumen engine's public interface that include/lumen_mi_renderer.hpp derives from and that
// examples/sandbox_driver.cpp calls, declared from scratch so that adapter + driver build and RUN on a machine without the reference
// tree (the GPU box): the abstract LumenRenderer with its payload structs, the ILumen* resource interfaces, ILumenScene / MeshInstance /
// Transform and the Camera.  Names, signatures and member meaning are those the adapter was written against
// (Lumen/src/Lumen/Renderer/LumenRenderer.h:29-219, ILumenResources.h:12-127, ModelLoading/ILumenScene.h:11-71, MeshInstance.h:14-112,
// Transform.h, Renderer/Camera.h — the build container compiles the same adapter and driver against those real headers:
// tests/test_cpu_host.py::test_adapter_and_driver_link_against_the_reference_sources_and_run_to_the_device_check).
// Behaviour is reduced to what the call sequence needs: a Transform is a world matrix, a Camera is position + rotation quaternion.
// Test scaffolding; not part of the product.
#pragma once
#include "glm_min.h"

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

class FrameSnapshot {};

// The vertex of an interleaved PrimitiveData, with the layout the reference's compilers give its `Vertex` (Shaders/CppCommon/ModelStructs.h:21-28): the members there
// are CUDA's float3 / float2 / float3 / float4, float2 is 8-byte and float4 16-byte aligned, hence 64 bytes with padding after the position and after the normal.
struct alignas(16) Vertex
{
    float m_Position[3];
    float m_PadAfterPosition;
    float m_UVCoord[2];
    float m_Normal[3];
    float m_PadAfterNormal[3];
    float m_Tangent[4];
};
static_assert(sizeof(Vertex) == 64, "Vertex must have the reference's size");

class Camera
{
public:
    Camera() = default;
    Camera(glm::vec3 position, glm::vec3 /*up*/) : m_Position(position) {}
    void SetPosition(glm::vec3 p) { m_Position = p; }
    void SetRotation(glm::quat q) { m_Rotation = q; }
    // columns: right, up, forward, position
    void GetMatrixData(glm::mat4& previous, glm::mat4& current)
    {
        current = glm::toMat4(m_Rotation);
        current[3] = glm::vec4(m_Position, 1.0f);
        if (!m_HavePrevious) { m_Previous = current; m_HavePrevious = true; }
        previous = m_Previous;
    }
    void UpdatePreviousFrameMatrix() { glm::mat4 p, c; GetMatrixData(p, c); m_Previous = c; }

private:
    glm::vec3 m_Position;
    glm::quat m_Rotation;
    glm::mat4 m_Previous;
    bool m_HavePrevious = false;
};

template <typename T, typename Byte>
class VectorView                      // a typed window over a byte vector (PrimitiveData's de-interleaved attribute streams)
{
public:
    VectorView() = default;
    explicit VectorView(std::vector<Byte>& bytes) : m_Bytes(&bytes) {}
    bool Empty() const { return !m_Bytes || m_Bytes->empty(); }
    size_t Size() const { return m_Bytes ? m_Bytes->size() / sizeof(T) : 0; }
    T& operator[](size_t i) { static T none{}; return m_Bytes && !m_Bytes->empty() ? reinterpret_cast<T*>(m_Bytes->data())[i] : none; }

private:
    std::vector<Byte>* m_Bytes = nullptr;
};

namespace Lumen
{
    class ILumenTexture { public: virtual ~ILumenTexture() = default; };
    class ILumenVolume { public: virtual ~ILumenVolume() = default; };

    class ILumenMaterial
    {
    public:
        virtual ~ILumenMaterial() = default;
        virtual void SetDiffuseColor(const glm::vec4&) = 0;
        virtual void SetDiffuseTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetEmission(const glm::vec3& = glm::vec3(0.f)) = 0;
        virtual void SetEmissiveTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetMetalRoughnessTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetNormalTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetClearCoatTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetClearCoatRoughnessTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetClearCoatFactor(float) = 0;
        virtual void SetClearCoatRoughnessFactor(float) = 0;
        virtual void SetLuminance(float) = 0;
        virtual void SetSheenFactor(float) = 0;
        virtual void SetSheenTintFactor(float) = 0;
        virtual void SetAnisotropic(float) = 0;
        virtual void SetTintTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetTintFactor(const glm::vec3&) = 0;
        virtual void SetTransmissionTexture(std::shared_ptr<ILumenTexture>) = 0;
        virtual void SetTransmissionFactor(float) = 0;
        virtual void SetTransmittanceFactor(const glm::vec3&) = 0;
        virtual void SetIndexOfRefraction(float) = 0;
        virtual void SetSpecularFactor(float) = 0;
        virtual void SetSpecularTintFactor(float) = 0;
        virtual void SetSubSurfaceFactor(float) = 0;
        virtual void SetMetallicFactor(float) = 0;
        virtual void SetRoughnessFactor(float) = 0;
        virtual float GetClearCoatFactor() = 0;
        virtual float GetClearCoatRoughnessFactor() = 0;
        virtual float GetLuminance() = 0;
        virtual float GetSheenFactor() = 0;
        virtual float GetSheenTintFactor() = 0;
        virtual float GetAnisotropic() = 0;
        virtual glm::vec3 GetTintFactor() = 0;
        virtual float GetTransmissionFactor() = 0;
        virtual glm::vec3 GetTransmittanceFactor() = 0;
        virtual float GetIndexOfRefraction() = 0;
        virtual float GetSpecularFactor() = 0;
        virtual float GetSpecularTintFactor() = 0;
        virtual float GetSubSurfaceFactor() = 0;
        virtual float GetMetallicFactor() = 0;
        virtual float GetRoughnessFactor() = 0;
        virtual glm::vec4 GetDiffuseColor() const = 0;
        virtual glm::vec3 GetEmissiveColor() const = 0;
        virtual ILumenTexture& GetDiffuseTexture() const = 0;
        virtual ILumenTexture& GetEmissiveTexture() const = 0;
    };

    class ILumenPrimitive
    {
    public:
        virtual ~ILumenPrimitive() = default;
        std::shared_ptr<ILumenMaterial> m_Material;
        bool m_ContainEmissive = false;
        unsigned int m_NumLights = 0;
    };

    class ILumenMesh
    {
    public:
        explicit ILumenMesh(std::vector<std::shared_ptr<ILumenPrimitive>>& primitives) : m_Primitives(primitives) {}
        virtual ~ILumenMesh() = default;
        std::vector<std::shared_ptr<ILumenPrimitive>> m_Primitives;
    };

    class Transform                   // reduced to the world matrix the renderer reads
    {
    public:
        Transform() : m_World(1.0f) {}
        Transform& operator=(const glm::mat4& m) { m_World = m; return *this; }
        glm::mat4 GetWorldTransformationMatrix() const { return m_World; }

    private:
        glm::mat4 m_World;
    };

    enum class EmissionMode { ENABLED, DISABLED, OVERRIDE };

    class MeshInstance
    {
    public:
        struct Emissiveness
        {
            Emissiveness(EmissionMode mode = EmissionMode::ENABLED, glm::vec3 radiance = glm::vec3(0.0f), float scale = 1.0f) : m_EmissionMode(mode), m_OverrideRadiance(radiance), m_Scale(scale) {}
            EmissionMode m_EmissionMode;
            glm::vec3 m_OverrideRadiance;
            float m_Scale;
        };
        virtual ~MeshInstance() = default;
        virtual void SetMesh(std::shared_ptr<ILumenMesh> mesh) { m_MeshRef = mesh; }
        virtual void SetEmissiveness(const Emissiveness& e) { m_EmissiveProperties = e; }
        virtual void SetOverrideMaterial(std::shared_ptr<ILumenMaterial> material) { m_OverrideMaterial = material; }
        Transform m_Transform;
        std::string m_Name;

    protected:
        Emissiveness m_EmissiveProperties;
        std::shared_ptr<ILumenMaterial> m_OverrideMaterial;
        std::shared_ptr<ILumenMesh> m_MeshRef;
    };

    class ILumenScene
    {
    public:
        ILumenScene(glm::vec3 cameraPosition = glm::vec3(0.f), glm::vec3 cameraUp = glm::vec3(0.f, 1.f, 0.f)) : m_Camera(std::make_unique<Camera>(cameraPosition, cameraUp)) {}
        virtual ~ILumenScene() = default;
        virtual MeshInstance* AddMesh() { m_MeshInstances.push_back(std::make_unique<MeshInstance>()); return m_MeshInstances.back().get(); }
        virtual void Clear() { m_MeshInstances.clear(); }
        std::vector<std::unique_ptr<MeshInstance>> m_MeshInstances;
        const std::unique_ptr<Camera> m_Camera;
        std::string m_Name;
    };

    // what a model file turns into (SceneManager.h:106-122), reduced to the pools the renderer-side loaders fill
    class SceneManager
    {
    public:
        struct GLTFResource
        {
            std::string m_Path;                                         // empty = no file was loaded
            std::vector<std::shared_ptr<ILumenMesh>> m_MeshPool;
            std::vector<std::shared_ptr<ILumenMaterial>> m_MaterialPool;
            std::vector<std::shared_ptr<ILumenScene>> m_Scenes;
        };
    };
}

struct FrameStats
{
    uint64_t m_Id = 0;
    std::map<std::string, uint64_t> m_Times;
};

class LumenRenderer
{
public:
    struct PrimitiveData
    {
        bool m_Interleaved = false;
        VectorView<glm::vec3, uint8_t> m_Positions;
        VectorView<glm::vec2, uint8_t> m_TexCoords;
        VectorView<glm::vec3, uint8_t> m_Normals;
        VectorView<glm::vec4, uint8_t> m_Tangents;
        std::vector<uint8_t> m_VertexBinary;
        std::vector<uint8_t> m_IndexBinary;
        size_t m_IndexSize = 4;
        std::shared_ptr<Lumen::ILumenMaterial> m_Material;
    };
    struct MaterialData
    {
        glm::vec4 m_DiffuseColor{1.f, 1.f, 1.f, 1.f};
        glm::vec3 m_EmissionVal{0.f, 0.f, 0.f};
        std::shared_ptr<Lumen::ILumenTexture> m_DiffuseTexture, m_NormalMap, m_MetallicRoughnessTexture, m_EmissiveTexture;
        std::shared_ptr<Lumen::ILumenTexture> m_TransmissionTexture, m_ClearCoatTexture, m_ClearCoatRoughnessTexture, m_TintTexture;
        float m_TransmissionFactor = 0.f, m_ClearCoatFactor = 0.f, m_ClearCoatRoughnessFactor = 0.f, m_IndexOfRefraction = 1.f, m_SpecularFactor = 0.f,
              m_SpecularTintFactor = 0.f, m_SubSurfaceFactor = 0.f, m_Luminance = 1.f, m_Anisotropic = 0.f, m_SheenFactor = 0.f, m_SheenTintFactor = 0.f,
              m_MetallicFactor = 1.f, m_RoughnessFactor = 1.f;
        glm::vec3 m_TintFactor{1.f, 1.f, 1.f};
        glm::vec3 m_Transmittance{1.f, 1.f, 1.f};
    };
    struct SceneData
    {
        SceneData() : m_CameraPosition(0.f, 0.f, 0.f), m_CameraUp(0.f, 1.f, 0.f) {}      // (a constructor instead of member initialisers: usable as a default argument below)
        glm::vec3 m_CameraPosition;
        glm::vec3 m_CameraUp;
    };

    LumenRenderer() = default;
    virtual ~LumenRenderer() = default;

    virtual void StartRendering() = 0;
    virtual void PerformDeferredOperations() {}
    virtual Lumen::SceneManager::GLTFResource OpenCustomFileFormat(const std::string&) { return {}; }      // LumenRenderer.h:154-155: renderer-side model cache
    virtual Lumen::SceneManager::GLTFResource CreateCustomFileFormat(const std::string&) { return {}; }
    virtual std::unique_ptr<Lumen::ILumenPrimitive> CreatePrimitive(PrimitiveData&) = 0;
    virtual std::shared_ptr<Lumen::ILumenMesh> CreateMesh(std::vector<std::shared_ptr<Lumen::ILumenPrimitive>>&) = 0;
    virtual std::shared_ptr<Lumen::ILumenTexture> CreateTexture(void* rgba8, uint32_t width, uint32_t height, bool normalize) = 0;
    virtual std::shared_ptr<Lumen::ILumenMaterial> CreateMaterial(const MaterialData&) = 0;
    virtual std::shared_ptr<Lumen::ILumenScene> CreateScene(SceneData = SceneData()) { return std::make_shared<Lumen::ILumenScene>(); }
    virtual std::shared_ptr<Lumen::ILumenVolume> CreateVolume(const std::string&) = 0;
    virtual void InitNGX() = 0;
    void CreateDefaultResources()     // three 1x1 textures, not normalised: white, default normal (128,128,255,0), white diffuse
    {
        uint8_t white[4] = {255, 255, 255, 255}, diffuse[4] = {255, 255, 255, 255}, normal[4] = {128, 128, 255, 0};
        m_DefaultWhiteTexture = CreateTexture(white, 1, 1, false);
        m_DefaultDiffuseTexture = CreateTexture(diffuse, 1, 1, false);
        m_DefaultNormalTexture = CreateTexture(normal, 1, 1, false);
    }
    virtual unsigned int GetOutputTexture() = 0;
    virtual std::vector<uint8_t> GetOutputTexturePixels(uint32_t& width, uint32_t& height) = 0;
    virtual void SetRenderResolution(glm::uvec2) = 0;
    virtual void SetOutputResolution(glm::uvec2) = 0;
    virtual void SetBlendMode(bool) = 0;
    virtual glm::uvec2 GetRenderResolution() = 0;
    virtual glm::uvec2 GetOutputResolution() = 0;
    virtual bool GetBlendMode() const = 0;
    virtual void BeginSnapshot() = 0;
    virtual std::unique_ptr<FrameSnapshot> EndSnapshot() = 0;
    FrameStats GetLastFrameStats() { std::lock_guard<std::mutex> lk(m_FrameStatsMutex); return m_LastFrameStats; }

    std::shared_ptr<Lumen::ILumenScene> m_Scene;

protected:
    FrameStats m_LastFrameStats;
    std::mutex m_FrameStatsMutex;

private:
    std::shared_ptr<Lumen::ILumenTexture> m_DefaultWhiteTexture, m_DefaultNormalTexture, m_DefaultDiffuseTexture;
};
