/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C API of the CPU restatement of the LumenPT wavefront path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (lumenrenderer_amd/, include/) never links, imports or executes anything under oracle/.
 *
 * Parity status: the reference as a whole cannot be built or run here (CUDA 10 + OptiX 7.1 + D3D11,
 * SURVEY.md §0 F10).  The header-only parts (RNG, material packing, Disney BSDF, Reservoir update / weight / reset,
 * CDF::Get / BinarySearch, make_color, binary16 conversion) ARE pinned against the reference's own headers through
 * tests/golden/ref_kat.npz, the kernel bodies of the ReSTIR chain, shading, primary rays, surface extraction, motion vectors and the light list through ref_kat5 / ref_kat6.npz
 * (line-range slices of the reference's .cu text run thread by thread), and the binary16 arithmetic of ShadeReservoirs / MergeOutputChannels — which decision D1 replaces by fp32 here —
 * through ref_kat7.npz (generator of all: oracle/ref_kat/, container-only).  Everything that
 * lives in OptiX programs or closed libraries (traversal, the texture unit's internal arithmetic — its published weight format IS followed —, thrust
 * sort/scan order, %smid bag choice) has no reference-side vectors:
 * for those stages this oracle is "parity unpinned" and is the definition the HIP path is held to.  Traversal: the hit rule is the published watertight test
 * (Woop, Benthin, Wald, JCGT 2013; lumen_oracle.cpp tri_hit) — like OptiX, it lets no ray pass between triangles that share an edge or a vertex
 * (tests/test_oracle_kat.py::test_oracle_hit_rule_is_watertight_at_shared_edges_and_vertices).
 */
#ifndef LUMEN_ORACLE_H
#define LUMEN_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* 25 floats in, see LumenRenderer.h:64-112 (MaterialData) for the meaning of each factor */
typedef struct orc_material_desc {
    float diffuse_color[4];
    float emission[3];
    int32_t tex_diffuse, tex_normal, tex_metal_rough, tex_emissive;      /* texture ids (>=0) */
    int32_t tex_transmission, tex_clearcoat, tex_clearcoat_rough, tex_tint;
    float transmission, clearcoat, clearcoat_roughness, ior, specular, specular_tint, subsurface,
          luminance, anisotropic, sheen, sheen_tint, metallic, roughness;
    float tint[3];
    float transmittance[3];
} orc_material_desc;

orc_ctx* orc_create(void);
void     orc_destroy(orc_ctx*);
void     orc_set_threads(orc_ctx*, int n);
/* bilinear filter: 0 (default) = the linear-filter rule the CUDA C Programming Guide publishes (weights in 1.8 fixed point), 1 = unquantised fp32 weights (decision D6) */
void     orc_set_tex_filter(orc_ctx*, int mode);

/* tex2D<float4>(texture, u, v) of the extraction (GPUExtractSurfaceData.cu:59-60,169-181) on n coordinates */
void orc_kat_tex2d(orc_ctx*, int texture, uint32_t n, const float* uv2, float* out4);
int  orc_add_texture(orc_ctx*, const uint8_t* rgba8, uint32_t w, uint32_t h, int srgb);
int  orc_add_material(orc_ctx*, const orc_material_desc*);
/* vertices: n * 12 floats (pos3 uv2 normal3 tangent4 = the 48-byte Vertex of ModelStructs.h:21-28) */
int  orc_add_primitive(orc_ctx*, const float* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, int material);
int  orc_add_mesh(orc_ctx*, const int* primitives, uint32_t n);
/* transform: row-major 4x4 world matrix. emission_mode: 0 ENABLED, 1 DISABLED, 2 OVERRIDE (MeshInstance.h:14-19) */
int  orc_add_instance(orc_ctx*, int mesh, const float transform[16], int emission_mode, const float override_radiance[3], float scale, int override_material);
void orc_set_instance_transform(orc_ctx*, int instance, const float transform[16]);
void orc_set_instance_emissiveness(orc_ctx*, int instance, int emission_mode, const float override_radiance[3], float scale);
void orc_set_instance_override_material(orc_ctx*, int instance, int material);
void orc_get_denoiser_inputs(orc_ctx*, float min_distance, float max_distance, float* depth, uint16_t* normal_roughness_half4, uint16_t* motion_half2);

/* camera: position + rotation matrix columns right/up/forward (Camera.cpp:122-140), vertical fov in degrees */
void orc_camera_vectors(const float right[3], const float up[3], const float forward[3], float fov_y_deg, float aspect, float out_uvw[9]);
void orc_motion_matrix(const float prev_cam_world_row_major[16], float fov_y_deg, float aspect, float out_row_major[16]);
void orc_set_camera(orc_ctx*, const float pos[3], const float right[3], const float up[3], const float forward[3], float fov_y_deg);
void orc_set_resolution(orc_ctx*, uint32_t w, uint32_t h);
void orc_set_depth(orc_ctx*, uint32_t depth);
void orc_set_blend(orc_ctx*, int blend);
/* render window [x0,x1) x [y0,y1) of the full image (tile sharding); default = whole image */
void orc_set_window(orc_ctx*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);

/* one TraceFrame(); returns 0, or 1 when the frame was skipped because the scene holds no lights */
int  orc_trace_frame(orc_ctx*);
void orc_get_radiance(orc_ctx*, float* rgba32f);            /* merged (blended) radiance, w*h*4 */
void orc_get_channel(orc_ctx*, int channel, float* rgba32f);/* 0 DIRECT 1 INDIRECT 2 SPECULAR 3 VOLUMETRIC */
void orc_get_output_pixels(orc_ctx*, uint8_t* rgba8);       /* sRGB RGBA8 as WriteToOutput produces */
/* per-frame counters: [0]=closest-hit rays, [1]=NEE shadow rays, [2]=ReSTIR shadow rays, [3]=lights,
 * [4..4+depth) = rays per wave */
void orc_get_stats(orc_ctx*, uint64_t* out, uint32_t n);

/* ---- unit-level entry points (known-answer tests) ---- */
uint32_t orc_wang_hash(uint32_t);
void  orc_random_floats(uint32_t seed, uint32_t n, float* out, uint32_t* states);
float orc_halton(uint32_t index, uint32_t base);
/* material "mat": 12 floats (color4 tint3 lum transmittance3 ior) + 11 parameter floats fed through the 8-bit setters */
void  orc_pack_material(const float mat[23], uint32_t params_out[3], float getters_out[11]);
void  orc_eval_bsdf(uint32_t n, const float* mat23, const float* N, const float* T, const float* wo, const float* wi, float* bsdf_pdf4);
void  orc_sample_bsdf(uint32_t n, const float* mat23, const float* N, const float* T, const float* wo, const float* r3, float* out8);
void  orc_det_math(uint32_t n, int fn, const float* x, const float* y, float* out);  /* fn: 0 sin 1 cos 2 log 3 exp 4 pow */
/* Reservoir::Update x k on a fresh reservoir (sample i carries id i + 1), then UpdateWeight, then Reset (ReSTIRData.h:115-178) */
void  orc_reservoir_sequence(uint32_t k, const float* w, const float* pdf, const uint32_t* seeds, float* weightSum, int64_t* count, int32_t* held, int32_t* took,
                             float* weight, float* afterReset3);
/* Resample / CombineBiased / CombineUnbiased (ReSTIRKernels.cu:1259-1325, :1200-1257, :1123-1198) on the row layout of
 * oracle/ref_kat/gen_kat4.cpp: surface(35) = position normal tangent incoming mat23; sample(14) = radiance normal position area
 * contribution solidAnglePdf; reservoir(17) = weightSum sampleCount weight sample(14) */
void  orc_resample(uint32_t n, const float* surf35, const float* sample14, float* out4);
void  orc_combine_biased(uint32_t n, uint32_t count, const float* surf35, const uint32_t* seeds, const float* res17, float* out17);
void  orc_combine_unbiased(uint32_t n, uint32_t count, const float* outsurf35, const uint32_t* seeds, const float* res17, const float* surfs35, float* out17);
/* CDF::Get / BinarySearch on a given prefix-sum array (ReSTIRData.h:230-306) */
void  orc_cdf_get(uint32_t n, const float* data, uint32_t m, const float* values, uint32_t* index, float* pdf);
/* make_color: sRGB8 of a linear colour (vendor/Include/Cuda/cuda/helpers.h:35-66) */
void  orc_make_color(uint32_t n, const float* rgb, uint8_t* rgba);
uint16_t orc_f32_to_f16(float);
float orc_f16_to_f32(uint16_t);

/* Known-answer entry points for the reference's KERNEL BODIES (rows of tests/golden/ref_kat5.npz; generator oracle/ref_kat/gen_kat5.cpp runs the reference's own
 * __global__ text thread by thread).  Arrays are 32-bit words as the rows store them: floats by bit pattern; flags, counts and indices as integers.
 * surface(40) = flags t position normal tangent incoming transport mat23;  reservoir(17) = weightSum sampleCount weight + sample(14);  light(16) = p0 p1 p2 normal radiance area. */
void orc_kat_light_weights(uint32_t n, const uint32_t* lights16, uint32_t* out);                                        /* CalculateLightWeightsInCDF ReSTIRKernels.cu:165-183 */
void orc_kat_primary_rays(uint32_t W, uint32_t H, uint32_t frameCount, const uint32_t* camUVWeye12, uint32_t* out11);  /* GeneratePrimaryRay GPUGeneratePrimRay.cu:28-82: x y origin dir contribution */
/* ShadeDirect GPUShadeDirect.cu:42-153 / ShadeIndirect GPUShadeIndirect.cu:7-146 on rows (x, y, seed, surface(40)); either output may be NULL.
 * direct12 = emitted origin direction maxDistance radiance channel; indirect10 = emitted origin direction contribution */
void orc_kat_shade(uint32_t n, uint32_t W, uint32_t H, const uint32_t* rows43, uint32_t nLights, const uint32_t* lights16, const uint32_t* cdf, uint32_t* direct12, uint32_t* indirect10);
/* scene-facing kernel bodies (rows of tests/golden/ref_kat6.npz) on a scene built through orc_add_*: ExtractSurfaceDataGpu GPUExtractSurfaceData.cu:8-228, GenerateMotionVector
 * MotionVectors.cu:8-55, FindEmissivesGpu GPUEmissiveLookup.cu:13-109, BuildLightDataBufferGPU GPUDataBufferKernels.cu:9-186; array shapes in lumen_oracle.cpp */
void orc_kat_extract(orc_ctx*, uint32_t n, const uint32_t* hits9, const uint32_t* rays9, uint32_t* out35);
void orc_kat_resolve(uint32_t n, const uint32_t* flags, const uint32_t* color4, uint32_t* out_half4);
void orc_kat_motion_vectors(uint32_t W, uint32_t H, const uint32_t* matrix16, const uint32_t* position_t4, uint32_t* out_half2);
uint32_t orc_kat_emissives(orc_ctx*, int primitive, uint8_t* flags);
uint32_t orc_kat_light_slots(orc_ctx*, uint32_t* out16, uint32_t capacity);
/* one ReSTIR::Run (Framework/ReSTIR.cpp:65-233; kernels ReSTIRKernels.cu:343-370,402-522,546-582,600-616,787-980,1015-1121,1407-1436) through the same restir_run that
 * renders, the visibility programs replaced by an occlusion mask per pass; see lumen_oracle.cpp for the array shapes */
void orc_kat_restir_frame(uint32_t W, uint32_t H, const uint32_t* surfCur40, const uint32_t* surfPrev40, const uint32_t* motionHalf2, uint32_t nLights, const uint32_t* lights16,
                          const uint32_t* cdf, uint32_t a_Seed, int currentIndex, const uint8_t* occ0, const uint8_t* occ1, uint32_t* res4, uint32_t* bags, uint32_t* stages,
                          uint32_t* rays, uint32_t* rayCounts, uint32_t* shadeFrom, uint32_t* direct);

/* ray-query seam (OptixWrapper::TraceRays): n rays (origin3, dir3) -> hits (instance, prim, u, v, t) */
void  orc_trace_closest(orc_ctx*, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax,
                        uint32_t* inst_prim, float* uvt, int use_bvh);
void  orc_trace_any(orc_ctx*, uint32_t n, const float* origins, const float* dirs, float tmin, const float* tmax, uint8_t* occluded, int use_bvh);
/* world-space triangle soup the tracer sees: returns count; fills 9 floats per triangle when out != NULL */
uint32_t orc_world_triangles(orc_ctx*, float* out);
/* sorted light list as ReSTIR sees it: 16 floats per light (p0 p1 p2 normal radiance area); returns count */
uint32_t orc_lights(orc_ctx*, float* out, float* cdf);
/* depth-0 surface data of the last frame: 8 float4 planes per pixel in the order documented in DESIGN.md */
void  orc_get_gbuffer(orc_ctx*, float* out);

#ifdef __cplusplus
}
#endif
#endif
