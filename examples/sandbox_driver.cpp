// sandbox_driver.cpp — the reference's own call sequence (Sandbox/src/Application.cpp:83-152, OutputLayer.cpp:119-168,882-896) driven through
// the reference-shaped adapter include/lumen_mi_renderer.hpp: construct the renderer where Sandbox constructs `LumenPT`, Init, default
// resources, textures / materials / primitives / meshes through the LumenRenderer virtuals, a scene with mesh instances, the scene's
// camera, StartRendering, the per-frame PerformDeferredOperations of the main loop, GetOutputTexturePixels -> a PPM.
//
//     sandbox_driver <scene file> <width> <height> <depth> <frames> <out.ppm>
//
// Two builds (tests/): against the reference tree's real headers + the handful of Lumen sources the interface needs (build container;
// without a GPU the run stops where lumen_mi_init reports that there is no device), and against the minimal interface headers of
// examples/sandbox_min/ on the GPU box, where the picture must equal the one examples/render_scene.c produces through the bare C ABI.
// Only the public interface of LumenRenderer / ILumenScene / MeshInstance / Camera is used — no adapter internals.
#include "lumen_mi_renderer.hpp"

#include <glm/gtc/quaternion.hpp>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

namespace
{
    struct Reader
    {
        std::ifstream f;
        explicit Reader(const char* path) : f(path, std::ios::binary) {}
        void Bytes(void* dst, size_t n) { f.read(static_cast<char*>(dst), static_cast<std::streamsize>(n)); if (!f) { std::fprintf(stderr, "scene file truncated\n"); std::exit(64); } }
        uint32_t U32() { uint32_t v; Bytes(&v, 4); return v; }
        float F32() { float v; Bytes(&v, 4); return v; }
    };
}

int main(int argc, char** argv)
{
    if (argc != 7) { std::fprintf(stderr, "usage: %s <scene file> <width> <height> <depth> <frames> <out.ppm>\n", argv[0]); return 64; }
    const unsigned width = static_cast<unsigned>(std::atoi(argv[2])), height = static_cast<unsigned>(std::atoi(argv[3])), depth = static_cast<unsigned>(std::atoi(argv[4]));
    const int frames = std::atoi(argv[5]);
    // a model file (.ollad / .gltf / .glb) goes through the renderer-side model cache exactly as SceneManager::LoadGLTF does (SceneManager.cpp:56-75); anything
    // else is this driver's own scene file.  SANDBOX_THREADED=1: StartRendering starts the render thread (the reference's behaviour); default: one frame per
    // PerformDeferredOperations call, so that the picture is that of exactly <frames> TraceFrames.
    const std::string scenePath = argv[1];
    const auto endsWith = [&](const char* e) { const size_t n = std::strlen(e); return scenePath.size() >= n && scenePath.compare(scenePath.size() - n, n, e) == 0; };
    const bool modelFile = endsWith(".ollad") || endsWith(".gltf") || endsWith(".glb");
    const bool threaded = std::getenv("SANDBOX_THREADED") != nullptr;
    float cam[13] = {0.f, 1.f, 3.4f, -1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, -1.f, 90.f};         // the Cornell box pose
    Reader in(modelFile ? (scenePath + ".cam").c_str() : argv[1]);
    if (modelFile) { if (in.f) in.Bytes(cam, sizeof cam); }                                          // optional side file: 13 floats (position, right, up, forward, fov)
    else {
        if (!in.f || in.U32() != 0x314D4C53u) { std::fprintf(stderr, "not a scene file\n"); return 64; }
        in.Bytes(cam, sizeof cam);
    }

    // Application.cpp:83-98
    auto renderer = std::make_shared<MI355X::Renderer>();
    MI355X::Renderer::Settings settings;
    settings.depth = depth;
    settings.renderResolution = {width, height};
    settings.outputResolution = {width, height};
    settings.blendOutput = true;
    settings.renderThread = threaded;
    renderer->Init(settings);
    renderer->CreateDefaultResources();
    // SANDBOX_GROUP=<rank>/<world>/<id file>: this process is one rank of a tile group (MI355X::Renderer::SetGroup; one process per GPU, rank 0 shows the stitched
    // frame).  Rank 0 obtains the communicator id from RCCL and leaves it in the file, the other ranks read it there.
    if (const char* spec = std::getenv("SANDBOX_GROUP")) {
        unsigned rank = 0, world = 0; char path[2048] = {0};
        if (std::sscanf(spec, "%u/%u/%2047s", &rank, &world, path) != 3 || rank >= world) { std::fprintf(stderr, "SANDBOX_GROUP=<rank>/<world>/<id file>\n"); return 64; }
        std::vector<uint8_t> id;
        if (rank == 0) {
            id = MI355X::Renderer::GroupUniqueId();
            const std::string tmp = std::string(path) + ".tmp";
            { std::ofstream w(tmp, std::ios::binary); w.write(reinterpret_cast<const char*>(id.data()), static_cast<std::streamsize>(id.size())); }
            std::rename(tmp.c_str(), path);
        } else {
            id.resize(LUMEN_MI_GROUP_ID_BYTES);
            for (int tries = 0; tries < 600; tries++) {
                std::ifstream r(path, std::ios::binary);
                if (r && r.read(reinterpret_cast<char*>(id.data()), static_cast<std::streamsize>(id.size()))) break;
                if (tries == 599) { std::fprintf(stderr, "rank %u: no communicator id in %s\n", rank, path); return 64; }
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        renderer->SetGroup(rank, world, id);
    }

    std::shared_ptr<Lumen::ILumenScene> scene;
    unsigned emissiveTriangles = 0;
    if (modelFile) {
        // SceneManager::LoadGLTF: first the optimised file, then its creation from the glTF (SceneManager.cpp:56-75)
        Lumen::SceneManager::GLTFResource res = renderer->OpenCustomFileFormat(scenePath);
        if (res.m_Path.empty()) res = renderer->CreateCustomFileFormat(scenePath);
        if (res.m_Path.empty() || res.m_Scenes.empty()) { std::fprintf(stderr, "the renderer could not open %s as a model file\n", scenePath.c_str()); return 65; }
        scene = res.m_Scenes[0];                                                                     // Application.cpp:143
        for (auto& mesh : res.m_MeshPool) for (auto& p : mesh->m_Primitives) emissiveTriangles += p->m_NumLights;
        std::printf("model file: %zu materials, %zu meshes, %zu instances\n", res.m_MaterialPool.size(), res.m_MeshPool.size(), scene->m_MeshInstances.size());
    } else {

    // what SceneManager does per glTF texture / material / primitive / mesh (SceneManager.cpp:277-541,704-822)
    std::vector<std::shared_ptr<Lumen::ILumenTexture>> textures(in.U32());
    for (auto& t : textures) {
        const uint32_t w = in.U32(), h = in.U32(), srgb = in.U32();
        std::vector<uint8_t> px(static_cast<size_t>(w) * h * 4);
        in.Bytes(px.data(), px.size());
        t = renderer->CreateTexture(px.data(), w, h, srgb != 0);
    }
    std::vector<std::shared_ptr<Lumen::ILumenMaterial>> materials(in.U32());
    for (auto& m : materials) {
        LumenRenderer::MaterialData d;
        uint32_t t[8]; float sc[13]; float v4[4], v3[3];
        in.Bytes(v4, 16); d.m_DiffuseColor = glm::vec4(v4[0], v4[1], v4[2], v4[3]);
        in.Bytes(v3, 12); d.m_EmissionVal = glm::vec3(v3[0], v3[1], v3[2]);
        in.Bytes(t, sizeof t); in.Bytes(sc, sizeof sc);
        for (uint32_t k : t) if (k >= textures.size()) { std::fprintf(stderr, "texture index out of range\n"); return 64; }
        d.m_DiffuseTexture = textures[t[0]]; d.m_NormalMap = textures[t[1]]; d.m_MetallicRoughnessTexture = textures[t[2]]; d.m_EmissiveTexture = textures[t[3]];
        d.m_TransmissionTexture = textures[t[4]]; d.m_ClearCoatTexture = textures[t[5]]; d.m_ClearCoatRoughnessTexture = textures[t[6]]; d.m_TintTexture = textures[t[7]];
        d.m_TransmissionFactor = sc[0]; d.m_ClearCoatFactor = sc[1]; d.m_ClearCoatRoughnessFactor = sc[2]; d.m_IndexOfRefraction = sc[3];
        d.m_SpecularFactor = sc[4]; d.m_SpecularTintFactor = sc[5]; d.m_SubSurfaceFactor = sc[6]; d.m_Luminance = sc[7]; d.m_Anisotropic = sc[8];
        d.m_SheenFactor = sc[9]; d.m_SheenTintFactor = sc[10]; d.m_MetallicFactor = sc[11]; d.m_RoughnessFactor = sc[12];
        in.Bytes(v3, 12); d.m_TintFactor = glm::vec3(v3[0], v3[1], v3[2]);
        in.Bytes(v3, 12); d.m_Transmittance = glm::vec3(v3[0], v3[1], v3[2]);
        m = renderer->CreateMaterial(d);
    }
    std::vector<std::shared_ptr<Lumen::ILumenPrimitive>> primitives(in.U32());
    for (auto& p : primitives) {
        const uint32_t m = in.U32(), nv = in.U32(), ni = in.U32();
        if (m >= materials.size()) { std::fprintf(stderr, "material index out of range\n"); return 64; }
        LumenRenderer::PrimitiveData d;
        d.m_Interleaved = true;                                   // `Vertex` records (ModelStructs.h:21-28), filled from the file's 12 floats per vertex
        d.m_VertexBinary.assign(static_cast<size_t>(nv) * sizeof(Vertex), 0);
        for (uint32_t v = 0; v < nv; v++) {
            float f[12]; in.Bytes(f, sizeof f);
            uint8_t* dst = d.m_VertexBinary.data() + static_cast<size_t>(v) * sizeof(Vertex);
            std::memcpy(dst + offsetof(Vertex, m_Position), f, 12); std::memcpy(dst + offsetof(Vertex, m_UVCoord), f + 3, 8);
            std::memcpy(dst + offsetof(Vertex, m_Normal), f + 5, 12); std::memcpy(dst + offsetof(Vertex, m_Tangent), f + 8, 16);
        }
        d.m_IndexBinary.resize(static_cast<size_t>(ni) * 4); in.Bytes(d.m_IndexBinary.data(), d.m_IndexBinary.size());
        d.m_IndexSize = 4;
        d.m_Material = materials[m];
        p = renderer->CreatePrimitive(d);                         // unique_ptr -> shared_ptr, as SceneManager stores them
        emissiveTriangles += p->m_NumLights;
    }
    std::vector<std::shared_ptr<Lumen::ILumenMesh>> meshes(in.U32());
    for (auto& mesh : meshes) {
        std::vector<std::shared_ptr<Lumen::ILumenPrimitive>> ps(in.U32());
        for (auto& p : ps) { const uint32_t pi = in.U32(); if (pi >= primitives.size()) { std::fprintf(stderr, "primitive index out of range\n"); return 64; } p = primitives[pi]; }
        mesh = renderer->CreateMesh(ps);
    }
    // Application.cpp:134-146: the scene, its instances, the camera
    scene = renderer->CreateScene();
    const uint32_t nInst = in.U32();
    for (uint32_t i = 0; i < nInst; i++) {
        const uint32_t m = in.U32();
        float xf[16], rad[4]; int32_t mode, overrideMaterial;
        in.Bytes(xf, sizeof xf); in.Bytes(&mode, 4); in.Bytes(rad, sizeof rad); in.Bytes(&overrideMaterial, 4);
        if (m >= meshes.size()) { std::fprintf(stderr, "mesh index out of range\n"); return 64; }
        Lumen::MeshInstance* inst = scene->AddMesh();
        inst->SetMesh(meshes[m]);
        glm::mat4 world;                                          // the file holds row-major matrices, glm is column-major
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) world[c][r] = xf[4 * r + c];
        inst->m_Transform = world;
        if (overrideMaterial >= 0) inst->SetOverrideMaterial(materials[static_cast<size_t>(overrideMaterial)]);
        inst->SetEmissiveness(Lumen::MeshInstance::Emissiveness(static_cast<Lumen::EmissionMode>(mode), glm::vec3(rad[0], rad[1], rad[2]), rad[3]));
    }
    }
    renderer->m_Scene = scene;
    const glm::mat3 basis(glm::vec3(cam[3], cam[4], cam[5]), glm::vec3(cam[6], cam[7], cam[8]), glm::vec3(cam[9], cam[10], cam[11]));   // columns right / up / forward (Camera.cpp:128-140)
    scene->m_Camera->SetRotation(glm::quat_cast(basis));
    scene->m_Camera->SetPosition(glm::vec3(cam[0], cam[1], cam[2]));

    // OutputLayer.cpp:492-495 pushes the blend mode every update; Application.cpp:152 starts the renderer; LumenApp::Run's loop
    // calls PerformDeferredOperations once per displayed frame (LumenApp.cpp:50-78 -> OutputLayer.cpp:119-168)
    renderer->SetBlendMode(true);
    renderer->StartRendering();
    if (!threaded) for (int k = 0; k < frames; k++) renderer->PerformDeferredOperations();
    else {
        // the render thread free-runs; the main loop keeps calling PerformDeferredOperations (LumenApp::Run) and, half way, moves the first instance the way the
        // tool UI does: the edit must reach the frames traced after it
        // (the thread may trace many frames between two iterations of this loop: the edit is made at the first iteration that sees half of the frames done, and the
        // loop then waits for two MORE frames, so that the edit is in the picture however fast the thread runs)
        bool moved = false;
        unsigned long long target = static_cast<unsigned long long>(frames);
        for (int spins = 0; spins < 200000; spins++) {
            renderer->PerformDeferredOperations();
            const unsigned long long id = renderer->GetLastFrameStats().m_Id;
            if (!moved && id >= static_cast<unsigned long long>(frames / 2) && !scene->m_MeshInstances.empty() && std::getenv("SANDBOX_MOVE")) {
                glm::mat4 w = scene->m_MeshInstances[0]->m_Transform.GetWorldTransformationMatrix();
                w[3].y += 0.25f;
                scene->m_MeshInstances[0]->m_Transform = w;
                moved = true;
                target = std::max(target, id + 2ull);
            }
            if (id >= target) break;
        }
        if (renderer->GetLastFrameStats().m_Id < target) { std::fprintf(stderr, "the render thread did not reach %d frames\n", frames); return 66; }
        // what the tracer sees while the thread is still running: the world-space triangle soup (a host-side product of the C ABI) — the edit above must be in it
        uint32_t nTris = 0;
        if (lumen_mi_get_world_triangles(renderer->Native(), nullptr, 0, &nTris) == LUMEN_MI_OK && nTris) {
            std::vector<float> tris(static_cast<size_t>(nTris) * 9);
            if (lumen_mi_get_world_triangles(renderer->Native(), tris.data(), nTris, &nTris) == LUMEN_MI_OK) {
                double sumY = 0.0;
                for (uint32_t k = 0; k < nTris * 3u; k++) sumY += tris[3u * k + 1u];
                std::printf("world triangles %u, sum of vertex heights %.6f\n", nTris, sumY);
            }
        }
    }

    uint32_t w = 0, h = 0;
    const std::vector<uint8_t> rgba = renderer->GetOutputTexturePixels(w, h);       // OutputLayer.cpp:882-896 (the screenshot path)
    std::FILE* o = std::fopen(argv[6], "wb");
    if (!o) { std::perror(argv[6]); return 64; }
    std::fprintf(o, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; i++) std::fwrite(rgba.data() + 4 * i, 1, 3, o);
    std::fclose(o);
    const FrameStats stats = renderer->GetLastFrameStats();
    std::printf("%ux%u, depth %u, %d frames, %u emissive triangles, frame id %llu\n", w, h, depth, frames, emissiveTriangles, static_cast<unsigned long long>(stats.m_Id));
    return 0;
}
