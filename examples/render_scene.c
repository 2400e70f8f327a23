/* render_scene.c — a caller of the C ABI (include/lumen_mi.h) in plain C, no Python and no reference headers:
 *
 *     render_scene <scene file> <width> <height> <depth> <frames> <out.ppm>
 *
 * reads a flat scene file (lumenrenderer_amd/scenes.py write_scene_file), replays it through the factories in the order the
 * reference's SceneManager would (textures, materials, primitives, meshes, scene, instances), renders `frames` blended
 * TraceFrames the way Sandbox's render loop does and writes the sRGB8 output (GetOutputTexturePixels) as a binary PPM.
 * Every failure prints lumen_mi_last_error() and exits with the status code of the call.
 *
 *     gcc -std=c99 -O2 -Iinclude examples/render_scene.c -o render_scene -Llumenrenderer_amd -llumen_mi -Wl,-rpath,$PWD/lumenrenderer_amd
 */
#include "lumen_mi.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != LUMEN_MI_OK) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, lumen_mi_last_error()); exit(rc_); } \
    } while (0)

static void rd(void* dst, size_t n, FILE* f) { if (n && fread(dst, 1, n, f) != n) { fprintf(stderr, "scene file truncated\n"); exit(64); } }
static uint32_t rd_u32(FILE* f) { uint32_t v; rd(&v, 4, f); return v; }
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(65); } return p; }

int main(int argc, char** argv)
{
    if (argc != 7) { fprintf(stderr, "usage: %s <scene file> <width> <height> <depth> <frames> <out.ppm>\n", argv[0]); return 64; }
    const uint32_t width = (uint32_t)atoi(argv[2]), height = (uint32_t)atoi(argv[3]), depth = (uint32_t)atoi(argv[4]);
    const int frames = atoi(argv[5]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 64; }
    if (rd_u32(f) != 0x314D4C53u) { fprintf(stderr, "not a scene file\n"); return 64; }
    float cam[13];
    rd(cam, sizeof cam, f);

    lumen_mi_renderer* r = NULL;
    CHECK(lumen_mi_create(&r));
    lumen_mi_settings s;
    memset(&s, 0, sizeof s);
    s.depth = depth; s.render_width = s.output_width = width; s.render_height = s.output_height = height; s.blend_output = 1; s.device = 0;
    CHECK(lumen_mi_init(r, &s));

    /* textures: the pixel memory is only borrowed during the call (SceneManager.cpp:737-743) */
    const uint32_t nTex = rd_u32(f);
    lumen_mi_handle* tex = (lumen_mi_handle*)xmalloc(nTex * sizeof *tex);
    for (uint32_t i = 0; i < nTex; i++) {
        const uint32_t w = rd_u32(f), h = rd_u32(f), srgb = rd_u32(f);
        void* px = xmalloc((size_t)w * h * 4);
        rd(px, (size_t)w * h * 4, f);
        CHECK(lumen_mi_create_texture(r, px, w, h, (int)srgb, &tex[i]));
        free(px);
    }
    const uint32_t nMat = rd_u32(f);
    lumen_mi_handle* mat = (lumen_mi_handle*)xmalloc(nMat * sizeof *mat);
    for (uint32_t i = 0; i < nMat; i++) {
        lumen_mi_material_data m;
        uint32_t t[8];
        float sc[13];
        memset(&m, 0, sizeof m);
        rd(m.diffuse_color, 16, f); rd(m.emission, 12, f); rd(t, sizeof t, f); rd(sc, sizeof sc, f); rd(m.tint_factor, 12, f); rd(m.transmittance, 12, f);
        for (int k = 0; k < 8; k++) if (t[k] >= nTex) { fprintf(stderr, "texture index out of range\n"); return 64; }
        m.diffuse_texture = tex[t[0]]; m.normal_map = tex[t[1]]; m.metallic_roughness_texture = tex[t[2]]; m.emissive_texture = tex[t[3]];
        m.transmission_texture = tex[t[4]]; m.clearcoat_texture = tex[t[5]]; m.clearcoat_roughness_texture = tex[t[6]]; m.tint_texture = tex[t[7]];
        m.transmission_factor = sc[0]; m.clearcoat_factor = sc[1]; m.clearcoat_roughness_factor = sc[2]; m.index_of_refraction = sc[3];
        m.specular_factor = sc[4]; m.specular_tint_factor = sc[5]; m.subsurface_factor = sc[6]; m.luminance = sc[7]; m.anisotropic = sc[8];
        m.sheen_factor = sc[9]; m.sheen_tint_factor = sc[10]; m.metallic_factor = sc[11]; m.roughness_factor = sc[12];
        CHECK(lumen_mi_create_material(r, &m, &mat[i]));
    }
    const uint32_t nPrim = rd_u32(f);
    lumen_mi_handle* prim = (lumen_mi_handle*)xmalloc(nPrim * sizeof *prim);
    uint32_t emissiveTriangles = 0;
    for (uint32_t i = 0; i < nPrim; i++) {
        lumen_mi_primitive_data p;
        memset(&p, 0, sizeof p);
        const uint32_t m = rd_u32(f);
        p.n_vertices = rd_u32(f); p.n_indices = rd_u32(f);
        if (m >= nMat) { fprintf(stderr, "material index out of range\n"); return 64; }
        void* v = xmalloc((size_t)p.n_vertices * 48);
        void* idx = xmalloc((size_t)p.n_indices * 4);
        rd(v, (size_t)p.n_vertices * 48, f); rd(idx, (size_t)p.n_indices * 4, f);
        p.interleaved = 1; p.vertex_binary = v; p.index_binary = idx; p.index_size = 4; p.material = mat[m];
        uint32_t lights = 0;
        CHECK(lumen_mi_create_primitive(r, &p, &prim[i], &lights));
        emissiveTriangles += lights;
        free(v); free(idx);
    }
    const uint32_t nMesh = rd_u32(f);
    lumen_mi_handle* mesh = (lumen_mi_handle*)xmalloc(nMesh * sizeof *mesh);
    for (uint32_t i = 0; i < nMesh; i++) {
        const uint32_t n = rd_u32(f);
        lumen_mi_handle* ps = (lumen_mi_handle*)xmalloc(n * sizeof *ps);
        for (uint32_t k = 0; k < n; k++) { const uint32_t pi = rd_u32(f); if (pi >= nPrim) { fprintf(stderr, "primitive index out of range\n"); return 64; } ps[k] = prim[pi]; }
        CHECK(lumen_mi_create_mesh(r, ps, n, &mesh[i]));
        free(ps);
    }
    lumen_mi_handle scene;
    CHECK(lumen_mi_create_scene(r, &scene));
    const uint32_t nInst = rd_u32(f);
    for (uint32_t i = 0; i < nInst; i++) {
        const uint32_t m = rd_u32(f);
        float xf[16], rad[4];
        int32_t mode, overrideMaterial;
        rd(xf, sizeof xf, f); rd(&mode, 4, f); rd(rad, sizeof rad, f); rd(&overrideMaterial, 4, f);
        if (m >= nMesh) { fprintf(stderr, "mesh index out of range\n"); return 64; }
        lumen_mi_handle inst;
        CHECK(lumen_mi_scene_add_mesh(r, scene, mesh[m], &inst));
        CHECK(lumen_mi_instance_set_transform(r, inst, xf));
        if (overrideMaterial >= 0) CHECK(lumen_mi_instance_set_override_material(r, inst, mat[overrideMaterial]));
        CHECK(lumen_mi_instance_set_emissiveness(r, inst, mode, rad, rad[3]));
    }
    fclose(f);
    CHECK(lumen_mi_set_scene(r, scene));
    CHECK(lumen_mi_camera_set(r, cam, cam + 3, cam + 6, cam + 9, cam[12]));

    /* Sandbox's loop: one TraceFrame per displayed frame, the renderer blends (SetBlendMode) */
    for (int k = 0; k < frames; k++) {
        const int rc = lumen_mi_trace_frame(r);
        if (rc == LUMEN_MI_NO_LIGHTS) { fprintf(stderr, "no emissive triangle in the scene: frame skipped like the reference does\n"); return rc; }
        if (rc != LUMEN_MI_OK) { fprintf(stderr, "lumen_mi_trace_frame: %d: %s\n", rc, lumen_mi_last_error()); return rc; }
    }
    uint8_t* rgba = (uint8_t*)xmalloc((size_t)width * height * 4);
    uint32_t w = 0, h = 0;
    CHECK(lumen_mi_get_output_pixels(r, rgba, (size_t)width * height * 4, &w, &h));
    uint64_t counters[8];
    CHECK(lumen_mi_get_counters(r, counters, 8));
    FILE* o = fopen(argv[6], "wb");
    if (!o) { perror(argv[6]); return 64; }
    fprintf(o, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) fwrite(rgba + 4 * i, 1, 3, o);
    fclose(o);
    printf("%ux%u, depth %u, %d frames, %u emissive triangles; last frame: %llu closest-hit rays, %llu shadow rays, %llu visibility rays\n", w, h, depth,
           frames, emissiveTriangles, (unsigned long long)counters[0], (unsigned long long)counters[1], (unsigned long long)counters[2]);
    free(rgba); free(tex); free(mat); free(prim); free(mesh);
    CHECK(lumen_mi_destroy(r));
    return 0;
}
