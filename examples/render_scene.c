/* render_scene.c — a caller of the C ABI (include/lumen_mi.h) in plain C, no Python and no reference headers:
 *
 *     render_scene <scene file> <width> <height> <depth> <frames> <out.ppm> [--ranks N --rank R --id-file PATH [--device D]]
 *
 * reads a flat scene file (lumenrenderer_amd/scenes.py write_scene_file), replays it through the factories in the order the
 * reference's SceneManager would (textures, materials, primitives, meshes, scene, instances), renders `frames` blended
 * TraceFrames the way Sandbox's render loop does and writes the sRGB8 output (GetOutputTexturePixels) as a binary PPM.
 * Every failure prints lumen_mi_last_error() and exits with the status code of the call.
 *
 * With --ranks N the program is ONE RANK of a tile group (include/lumen_mi.h "tile groups"): start it N times, one process per GPU (--device, default = the rank), with
 * the same --id-file.  Rank 0 asks RCCL for the communicator id and writes it to that file, the other ranks wait for it; every rank renders its tile + halo, the tiles are
 * gathered on rank 0 over RCCL (double-buffered, on a stream of their own: nothing waits between the frames), and rank 0 writes the stitched frame — the fp32 radiance
 * through the sRGB transfer function below — as the PPM.  `--ranks 1` runs the same code on one GPU, over RCCL as well.
 *
 *     gcc -std=c99 -O2 -Iinclude examples/render_scene.c -o render_scene -Llumenrenderer_amd -llumen_mi -lm -Wl,-rpath,$PWD/lumenrenderer_amd
 */
#define _POSIX_C_SOURCE 199309L      /* nanosleep */
#include "lumen_mi.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != LUMEN_MI_OK) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, lumen_mi_last_error()); exit(rc_); } \
    } while (0)

static void rd(void* dst, size_t n, FILE* f) { if (n && fread(dst, 1, n, f) != n) { fprintf(stderr, "scene file truncated\n"); exit(64); } }
static uint32_t rd_u32(FILE* f) { uint32_t v; rd(&v, 4, f); return v; }
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(65); } return p; }

/* linear -> sRGB8 of the group path's PPM: the IEC 61966-2-1 transfer function, 256 levels, clamped (the single-GPU path reads the renderer's own sRGB8 output instead) */
static uint8_t srgb8(float c)
{
    const float s = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    const float q = (s < 0.f ? 0.f : s > 1.f ? 1.f : s) * 256.f;
    return (uint8_t)(q > 255.f ? 255.f : q);
}

int main(int argc, char** argv)
{
    uint32_t ranks = 0, rank = 0;
    int device = -1;
    const char* idFile = NULL;
    for (int a = 7; a + 1 < argc; a += 2) {
        if (!strcmp(argv[a], "--ranks")) ranks = (uint32_t)atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "--rank")) rank = (uint32_t)atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "--id-file")) idFile = argv[a + 1];
        else if (!strcmp(argv[a], "--device")) device = atoi(argv[a + 1]);
        else { fprintf(stderr, "unknown option %s\n", argv[a]); return 64; }
    }
    if (argc < 7 || (argc - 7) % 2 || (ranks && (rank >= ranks || !idFile))) {
        fprintf(stderr, "usage: %s <scene file> <width> <height> <depth> <frames> <out.ppm> [--ranks N --rank R --id-file PATH [--device D]]\n", argv[0]); return 64;
    }
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
    s.depth = depth; s.render_width = s.output_width = width; s.render_height = s.output_height = height; s.blend_output = 1; s.device = device >= 0 ? device : (int)rank;
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

    if (ranks) {
        /* one rank of a tile group: id from rank 0 through the file, then frames + gathers back to back; rank 0 writes the stitched frame */
        uint8_t id[LUMEN_MI_GROUP_ID_BYTES];
        if (rank == 0) {
            CHECK(lumen_mi_group_unique_id(id));
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", idFile);
            FILE* w = fopen(tmp, "wb");
            if (!w || fwrite(id, 1, sizeof id, w) != sizeof id || fclose(w) || rename(tmp, idFile)) { perror(idFile); return 64; }
        } else {
            int tries = 0;
            FILE* w = NULL;
            const struct timespec tenth = {0, 100000000L};
            while (!(w = fopen(idFile, "rb")) && tries++ < 600) nanosleep(&tenth, NULL);
            if (!w || fread(id, 1, sizeof id, w) != sizeof id) { fprintf(stderr, "rank %u: no communicator id in %s after 60 s\n", rank, idFile); return 64; }
            fclose(w);
        }
        lumen_mi_group* g = NULL;
        CHECK(lumen_mi_group_create(r, rank, ranks, id, NULL, &g));
        float ms = 0.f;
        CHECK(lumen_mi_group_self_test(g, &ms));
        lumen_mi_tile_plan plan;
        CHECK(lumen_mi_group_get_plan(g, &plan));
        for (int k = 0; k < frames; k++) {
            const int rc = lumen_mi_group_trace_frame(g);
            if (rc != LUMEN_MI_OK) { fprintf(stderr, "rank %u: lumen_mi_group_trace_frame: %d: %s\n", rank, rc, lumen_mi_last_error()); return rc; }
            CHECK(lumen_mi_group_gather(g));
        }
        CHECK(lumen_mi_group_synchronize(g));
        uint64_t gathers = 0;
        float gatherMs = 0.f;
        CHECK(lumen_mi_group_get_stats(g, &gathers, &gatherMs));
        printf("rank %u of %u: grid %ux%u, tile [%u,%u)x[%u,%u), window [%u,%u)x[%u,%u), self-test %.1f ms, %llu gathers\n", rank, ranks, plan.cols, plan.rows,
               plan.tile[0], plan.tile[2], plan.tile[1], plan.tile[3], plan.window[0], plan.window[2], plan.window[1], plan.window[3], ms, (unsigned long long)gathers);
        if (rank == 0) {
            float* frame = (float*)xmalloc((size_t)width * height * 16);
            CHECK(lumen_mi_group_get_frame(g, frame, (size_t)width * height * 16));
            FILE* o = fopen(argv[6], "wb");
            if (!o) { perror(argv[6]); return 64; }
            fprintf(o, "P6\n%u %u\n255\n", width, height);
            for (size_t i = 0; i < (size_t)width * height; i++) { const uint8_t px[3] = {srgb8(frame[4 * i]), srgb8(frame[4 * i + 1]), srgb8(frame[4 * i + 2])}; fwrite(px, 1, 3, o); }
            fclose(o);
            free(frame);
            printf("%ux%u, depth %u, %d frames, %u emissive triangles, stitched from %u tile(s)\n", width, height, depth, frames, emissiveTriangles, ranks);
        }
        CHECK(lumen_mi_group_destroy(g));
        free(tex); free(mat); free(prim); free(mesh);
        CHECK(lumen_mi_destroy(r));
        return 0;
    }

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
