/* lumen_mi.h — C ABI of the MI355X-native wavefront path tracer ("liblumen_mi.so").
 *
 * The reference has no C ABI: its renderer is the C++ abstract class LumenRenderer, statically linked and chosen at
 * compile time (Lumen/src/Lumen/Renderer/LumenRenderer.h:37-219, LumenPT/src/LumenPT.h:8-20,
 * Sandbox/src/Application.cpp:81-99).  Each entry point below replaces one member of that class (or of the
 * objects it hands out) and cites it; include/lumen_mi_renderer.hpp is the header-only C++ adapter that gives
 * the reference's own class shape back on top of these calls.
 *
 * Conventions: plain pointers and sizes only; handles are opaque 64-bit integers; every call returns an int
 * status (0 = LUMEN_MI_OK) and lumen_mi_last_error() describes the last failure of the calling thread.  Pixel,
 * vertex and index memory is borrowed for the duration of the call only (as in the reference,
 * SceneManager.cpp:737-743).  Matrices are row-major float[16].  Nothing here includes HIP headers.
 */
#ifndef LUMEN_MI_H
#define LUMEN_MI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lumen_mi_renderer lumen_mi_renderer;
typedef uint64_t lumen_mi_handle;

enum {
    LUMEN_MI_OK = 0,
    LUMEN_MI_ERR_INVALID = 1,       /* bad argument / handle */
    LUMEN_MI_ERR_DEVICE = 2,        /* HIP error (no GPU, out of memory, launch failure) */
    LUMEN_MI_ERR_STATE = 3,         /* call order (e.g. trace before init / without a scene) */
    LUMEN_MI_NO_LIGHTS = 4          /* frame skipped: the scene holds no emissive triangle (WaveFrontRenderer.cpp:456-464) */
};

/* WaveFrontSettings (LumenPT/src/Framework/WaveFrontRenderer.h:31-48); shader paths have no meaning here */
typedef struct lumen_mi_settings {
    uint32_t depth;                 /* maximum path depth (Sandbox: 5) */
    uint32_t render_width, render_height;
    uint32_t output_width, output_height;
    int32_t  blend_output;
    int32_t  device;                /* HIP device ordinal of this process */
} lumen_mi_settings;

/* LumenRenderer::MaterialData (LumenRenderer.h:64-112); textures are handles from lumen_mi_create_texture (0 = none) */
typedef struct lumen_mi_material_data {
    float diffuse_color[4];
    float emission[3];
    lumen_mi_handle diffuse_texture, normal_map, metallic_roughness_texture, emissive_texture;
    lumen_mi_handle transmission_texture, clearcoat_texture, clearcoat_roughness_texture, tint_texture;
    float transmission_factor, clearcoat_factor, clearcoat_roughness_factor, index_of_refraction;
    float specular_factor, specular_tint_factor, subsurface_factor, luminance, anisotropic;
    float sheen_factor, sheen_tint_factor, metallic_factor, roughness_factor;
    float tint_factor[3];
    float transmittance[3];
} lumen_mi_material_data;

/* LumenRenderer::PrimitiveData (LumenRenderer.h:44-61).  Either interleaved vertices or separate attribute arrays (any of uv/normal/tangent may be NULL).
 * interleaved: 0 = separate arrays; 1 = 48-byte vertices, tightly packed floats (position 3, uv 2, normal 3, tangent 4);
 * 2 = the reference's `Vertex` (Shaders/CppCommon/ModelStructs.h:21-28) AS ITS COMPILERS LAY IT OUT: the members are CUDA vector types (`LUMEN` is defined nowhere in
 * the reference's build), float2 is 8-byte and float4 16-byte aligned, so a Vertex is 64 bytes — position at byte 0, uv at 16, normal at 24, tangent at 48 — which is
 * what an interleaved PrimitiveData from the reference's model converter (and every .ollad file) carries.  The adapter passes 2. */
enum { LUMEN_MI_VERTICES_SEPARATE = 0, LUMEN_MI_VERTICES_PACKED48 = 1, LUMEN_MI_VERTICES_REFERENCE64 = 2 };
typedef struct lumen_mi_primitive_data {
    int32_t interleaved;
    const void* vertex_binary;      /* interleaved: n_vertices * 48 (layout 1) or * 64 (layout 2) bytes */
    const float* positions;         /* 3 floats per vertex */
    const float* tex_coords;        /* 2 */
    const float* normals;           /* 3 */
    const float* tangents;          /* 4 */
    uint32_t n_vertices;
    const void* index_binary;
    uint32_t n_indices;
    uint32_t index_size;            /* 2 or 4 bytes */
    lumen_mi_handle material;
} lumen_mi_primitive_data;

/* Lumen::EmissionMode (MeshInstance.h:14-19) */
enum { LUMEN_MI_EMISSION_ENABLED = 0, LUMEN_MI_EMISSION_DISABLED = 1, LUMEN_MI_EMISSION_OVERRIDE = 2 };

/* ---- lifetime: `new WaveFrontRenderer` + WaveFrontRenderer::Init (WaveFrontRenderer.h:86, .cpp:70-322), dtor (.cpp:1360-1371) */
int lumen_mi_create(lumen_mi_renderer** out);
int lumen_mi_init(lumen_mi_renderer*, const lumen_mi_settings*);
int lumen_mi_destroy(lumen_mi_renderer*);
const char* lumen_mi_last_error(void);
/* stream every kernel of this renderer is enqueued on (hipStream_t passed as void*; NULL = default stream) */
int lumen_mi_set_stream(lumen_mi_renderer*, void* hip_stream);

/* ---- resource factories */
int lumen_mi_create_texture(lumen_mi_renderer*, const void* rgba8, uint32_t width, uint32_t height, int normalize, lumen_mi_handle* out);      /* CreateTexture  LumenRenderer.h:161 */
int lumen_mi_create_material(lumen_mi_renderer*, const lumen_mi_material_data*, lumen_mi_handle* out);                                     /* CreateMaterial LumenRenderer.h:164 */
int lumen_mi_update_material(lumen_mi_renderer*, lumen_mi_handle material, const lumen_mi_material_data*);                                    /* ILumenMaterial setters, ILumenResources.h:23-55 */
int lumen_mi_create_default_resources(lumen_mi_renderer*, lumen_mi_handle* white, lumen_mi_handle* normal, lumen_mi_handle* diffuse);      /* CreateDefaultResources LumenRenderer.cpp:50-58 */
int lumen_mi_create_primitive(lumen_mi_renderer*, const lumen_mi_primitive_data*, lumen_mi_handle* out, uint32_t* num_lights);               /* CreatePrimitive LumenRenderer.h:157; ILumenPrimitive::m_NumLights */
int lumen_mi_create_mesh(lumen_mi_renderer*, const lumen_mi_handle* primitives, uint32_t n, lumen_mi_handle* out);                          /* CreateMesh     LumenRenderer.h:159 */
int lumen_mi_create_scene(lumen_mi_renderer*, lumen_mi_handle* out);                                                                       /* CreateScene    LumenRenderer.h:166 */
int lumen_mi_set_scene(lumen_mi_renderer*, lumen_mi_handle scene);                                                                         /* m_Scene        LumenRenderer.h:201 */

/* ---- ILumenScene / MeshInstance (ILumenScene.h:48-67, MeshInstance.h:22-112) */
int lumen_mi_scene_add_mesh(lumen_mi_renderer*, lumen_mi_handle scene, lumen_mi_handle mesh, lumen_mi_handle* instance_out);               /* AddMesh()->SetMesh() */
int lumen_mi_scene_clear(lumen_mi_renderer*, lumen_mi_handle scene);                                                                       /* Clear(): the scene's instance handles are released (later use: LUMEN_MI_ERR_INVALID) */
int lumen_mi_instance_set_transform(lumen_mi_renderer*, lumen_mi_handle instance, const float world_matrix[16]);                           /* m_Transform */
int lumen_mi_instance_set_emissiveness(lumen_mi_renderer*, lumen_mi_handle instance, int mode, const float override_radiance[3], float scale); /* SetEmissiveness */
int lumen_mi_instance_set_override_material(lumen_mi_renderer*, lumen_mi_handle instance, lumen_mi_handle material);                       /* SetOverrideMaterial */

/* ---- camera (Lumen/src/Lumen/Renderer/Camera.h:14-64): position + rotation matrix columns right/up/forward + vertical FOV */
int lumen_mi_camera_set(lumen_mi_renderer*, const float position[3], const float right[3], const float up[3], const float forward[3], float fov_y_degrees);

/* ---- settings (LumenRenderer.h:178-196) */
int lumen_mi_set_render_resolution(lumen_mi_renderer*, uint32_t w, uint32_t h);     /* also forces the output resolution (WaveFrontRenderer.cpp:352) */
int lumen_mi_set_output_resolution(lumen_mi_renderer*, uint32_t w, uint32_t h);
int lumen_mi_get_render_resolution(lumen_mi_renderer*, uint32_t* w, uint32_t* h);
int lumen_mi_get_output_resolution(lumen_mi_renderer*, uint32_t* w, uint32_t* h);
int lumen_mi_set_blend_mode(lumen_mi_renderer*, int blend);                          /* resets the blend counter when enabling (WaveFrontRenderer.cpp:377-381) */
int lumen_mi_get_blend_mode(lumen_mi_renderer*, int* blend);
int lumen_mi_set_depth(lumen_mi_renderer*, uint32_t depth);                          /* WaveFrontSettings::depth */

/* ---- rendering */
int lumen_mi_trace_frame(lumen_mi_renderer*);           /* body of WaveFrontRenderer::TraceFrame (.cpp:435-1089); blocking */
int lumen_mi_trace_frame_async(lumen_mi_renderer*);     /* enqueue only; lumen_mi_synchronize() or any readback completes it */
int lumen_mi_synchronize(lumen_mi_renderer*);
int lumen_mi_start_rendering(lumen_mi_renderer*);       /* StartRendering LumenRenderer.h:151: render thread looping TraceFrame (.cpp:1109-1117) */
int lumen_mi_stop_rendering(lumen_mi_renderer*);
int lumen_mi_perform_deferred_operations(lumen_mi_renderer*);   /* PerformDeferredOperations LumenRenderer.h:152 (nothing is deferred here) */

/* ---- readback */
int lumen_mi_get_output_pixels(lumen_mi_renderer*, uint8_t* rgba8, size_t capacity_bytes, uint32_t* w, uint32_t* h);   /* GetOutputTexturePixels LumenRenderer.h:176 */
int lumen_mi_get_radiance(lumen_mi_renderer*, float* rgba32f, size_t capacity_bytes);     /* merged fp32 radiance of the render window (no reference equivalent; parity checks) */
/* The same radiance as the reference STORES it: binary16 RGBA, round to nearest even (its pixel buffers are half4 surfaces,
 * GPUMergeOutputChannels.cu:5-88, Half4.h:9-96).  8 bytes per pixel of the render window. */
int lumen_mi_get_radiance_half4(lumen_mi_renderer*, uint16_t* rgba16f, size_t capacity_bytes);
int lumen_mi_copy_radiance_device(lumen_mi_renderer*, void* device_rgba32f);              /* same, device-to-device on the renderer's stream (RCCL gather source) */
/* Tile gather (multi-GPU; no reference equivalent): the rectangle [x0, x1) x [y0, y1) of the IMAGE, which must lie inside the render window, copied
 * out of the merged radiance into a device image of `dst_pitch` RGBA32F pixels per row (dst = the rectangle's first pixel); and a w x h rectangle between two pitched
 * device images (a gathered tile into the assembled frame).  Both run on the renderer's stream as a kernel of this library. */
int lumen_mi_copy_radiance_rect_device(lumen_mi_renderer*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* device_dst, uint32_t dst_pitch);
int lumen_mi_copy_rect_device(lumen_mi_renderer*, void* device_dst, uint32_t dst_pitch, const void* device_src, uint32_t src_pitch, uint32_t w, uint32_t h);
int lumen_mi_get_channel(lumen_mi_renderer*, int channel, float* rgba32f, size_t capacity_bytes);   /* 0 DIRECT, 1 INDIRECT */
int lumen_mi_get_gbuffer(lumen_mi_renderer*, float* planes8x4, size_t capacity_bytes);   /* depth-0 surface data of the last frame, pixel-major [n][8][4] */

/* denoiser / upscaler inputs of the last frame (ExtractNRD_DLSSdata GPUExtractNRD_DLSSdata.cu:6-89, ExtractDepthData
 * GPUExtractDepthData.cu:6-72, GenerateMotionVector MotionVectors.cu:8-55): depth normalised to [min,max] render distance
 * (Camera.h:60 default {0.1, 1000}) as fp32, normal.xyz + roughness as half4, motion vector as half2; any pointer may be NULL */
int lumen_mi_get_denoiser_inputs(lumen_mi_renderer*, float min_distance, float max_distance, float* depth, uint16_t* normal_roughness_half4, uint16_t* motion_half2);

/* FrameStats (LumenRenderer.h:29-34, GetLastFrameStats :203): key/value pairs in microseconds under the reference's key names; plus the key "Frames Traced" =
 * the number of TraceFrames enqueued since the renderer was created (what FrameStats::m_Id counts; readable while the render thread runs) */
int lumen_mi_get_frame_stat(lumen_mi_renderer*, const char* key, uint64_t* microseconds);
/* counters of the last completed frame: [0] closest-hit rays, [1] NEE shadow rays, [2] ReSTIR shadow rays, [3] lights,
 * [4..4+depth) rays per wave, [20] BVH nodes visited in binary-node equivalents (= [22] / 2), [21] triangles tested,
 * [22] child boxes slab-tested by the 4-wide traversal, [24..40) histogram of per-ray traversal steps in log2 buckets,
 * [40] the longest per-ray traversal in steps, [41]/[42] active lanes / lane slots over all node steps, [43]/[44] the same
 * over all triangle tests, [45]/[46] traversal-stack pushes into LDS / into the global spill area
 * ([20]..[46] only in the instrumented build), [48]/[49] ReSTIR visibility rays of pass 1 / pass 2, [50] GPU refits and
 * [51] instance-level tree assemblies since the renderer was created, [52] depth-0 surfaces outside the contracted ReSTIR evaluation seen (flag),
 * [53] can any material produce one, [56] full tree builds done on the device since the renderer was created (tuning key gpu_build), [54] / [55] lazy reuse (tuning key lazy_reuse): deferred executions of the history passes / reservoir entries whose
 * sample count was completed instead, since the reservoirs were last reset */
int lumen_mi_get_counters(lumen_mi_renderer*, uint64_t* out, uint32_t n);
/* the same counters SUMMED over every TraceFrame since the renderer was created or since the last call with reset != 0 (accumulated on the
 * device by the frame's last kernel, no read-back between frames): [0] closest-hit rays, [1] NEE shadow rays, [2] ReSTIR shadow rays,
 * [3] number of TraceFrames summed, [4..4+depth) rays per wave, [48]/[49] ReSTIR visibility rays of pass 1 / 2.  What bench.py divides by
 * the wall time of the frames it timed (the reference's counters: WaveFrontRenderer.cpp:700-703,815,837; ReSTIRKernels.cu:543). */
int lumen_mi_get_counter_totals(lumen_mi_renderer*, uint64_t* out, uint32_t n, int reset);
/* device time of one kernel class, summed over every frame traced since timing was enabled, measured with HIP events
 * on the renderer's stream; `launches` = number of timed launches (class 4: number of frames).
 * which: 0 closest-hit traversal, 1 shadow traversal, 2 extract+shade, 3 ReSTIR (all passes), 4 whole frame, 5 path tail (the deep waves in one launch).
 * enable: 0 off, 1 every class, 2 classes 0 and 4 only (fewer events in the streams: what a throughput measurement wants) */
int lumen_mi_get_kernel_time(lumen_mi_renderer*, int which, float* milliseconds, uint32_t* launches);
int lumen_mi_enable_kernel_timing(lumen_mi_renderer*, int enable);
int lumen_mi_set_instrumented(lumen_mi_renderer*, int enable);   /* use the node/triangle counting traversal kernels */
/* Scheduling knobs (no reference equivalent).  None of them changes a result, except the three arithmetic keys at the end of this list ("fast_resample", "fast_shade":
 * within the stated tolerance) and the test-only value 2 of "lazy_reuse".  Keys: "tail_below" (waves expected to hold fewer rays run
 * as one path-tail launch; 0 = off, -1 = automatic), "tail_lanes" (paths per wavefront in that launch, 1..64; 0 or less = automatic), "tail_pair" (the NEE
 * shadow ray a path emits at one depth is traced by a partner lane beside the path's closest-hit query of the next depth: 1 on, 0 off, -1 automatic), "single_stream" (1 = no stream
 * overlap, no frame pipelining), "pick_ahead" (ReSTIR candidate generation of the next frame on its own stream: 1 on, 0 off,
 * -1 automatic), "refill" / "refill_visibility" (lane-refill thresholds of the queue traversal), "shadow_on_wave" (NEE shadow rays on
 * the wave stream), "wave_streams" (1, default: one wave stream, NEE shadows and the path tail on a stream beside it; 2: the path-tracing launches of
 * even / odd frames alternate between those two streams, so that the wave chains of consecutive frames overlap), "fuzz" (test aid: a seed != 0 inserts idle launches of random length in front of the kernels of a frame; the image
 * must not change), "assemble" (1, default: after the first build a topology edit — an instance added or removed — assembles cached
 * per-mesh trees behind a small top tree and refits on the GPU; 0: full host SAH rebuild), "packet_primary" / "packet_visibility" (the primary
 * wave / the ReSTIR visibility rays are traced as wavefront packets — one shared traversal stack per 64 coherent rays: 1 on, 0 off, -1
 * automatic: on when the window has more than 4 pixels per scene triangle; default -1 for the primary wave, 0 for the visibility rays, where packets measured slower), "fuse_primary" (1: the packet kernel of the primary wave generates its rays itself instead of reading a plane written by a launch before it; default 0, no gain measured),
 * "lazy_reuse" (the history-building ReSTIR passes of a frame — both spatial reuse passes and CombineReservoirBuffers,
 * ReSTIR.cpp:181-233 — are launched with the NEXT frame and run only if their result can still be read, i.e. when the reservoir swap chain has turned; if it has
 * not — every frame of an even path depth, WaveFrontRenderer.cpp:827 — only the sample counts of the entries that outlive the next candidate pick are
 * completed.  Images, counters and exported history counts equal those of launching the passes with their frame; 1 on, 0 off, -1 (default) automatic: on at
 * even path depths; 2 = TEST ONLY: on WITHOUT the count completion, wrong on purpose, for the test that shows the completion is observable — reachable through this key
 * only, the environment variable LUMEN_MI_LAZY_REUSE is clamped to -1 .. 1; DESIGN.md "Lazy reuse"), "gpu_build" (1: a scene's full tree build runs on the device — Morton sort, radix tree, collapse to the 4-wide layout, boxes and packets by the refit kernels
 * (csrc/bvh_gpu.hip; the reference builds its acceleration structures on the GPU too, OptixWrapper.cpp:46-78); 0, default: the host's binned-SAH builder, a better tree that takes
 * ten times as long.  Hit records do not depend on the tree: identical images either way), "tail_repack" (1: the path tail repacks its live paths across the block's
 * wavefronts after every depth; identical image, measured slower, default 0), "spatial_lds" (1: in the fast mode the first spatial reuse pass stages
 * the probes of a 32 x 32 pixel tile + its 30-pixel reach in LDS, 132 KB per block; 2: the same with the ordinary 16 x 16 tile, 92 KB per block; identical image; default 0: either evicts the other streams' kernels, measured slower on the frame),
 * "fast_resample" (arithmetic mode of the ReSTIR passes: hardware rcp / rsq / sqrt and the contracted target function; radiance within 1e-3 relative L2 of the exact mode,
 * 1e-8 measured), "fast_shade" (on top of it: the NEE contribution at depth >= 1 with hardware rcp / sqrt: changes the last bits of INDIRECT radiance and, within
 * rounding of two thresholds, whether a shadow ray is emitted; never which path continues), "pick_wide" (light lists of 513 .. 1 984 triangles: the candidate pick runs as
 * 1024-thread blocks, four tiles around ONE light table in LDS, instead of gathering the lights from memory: 0 never, 1 in the fast mode (default; the exact instantiation is
 * slower that way), 2 in both modes; identical image), "trace_blocks_main" / "trace_blocks_vis" / "trace_blocks_aux" (blocks per CU, 1 .. 8, of the persistent
 * traversal launches: primary rays / the two ReSTIR visibility passes / waves >= 1 and shadow rays; main and vis: 0, default = chosen per frame — half the grid where the launch
 * runs beside the history passes and can afford it, csrc/frame.cpp; identical image), "fuse_combine" (1, default: on eager frames the second spatial reuse pass ends with the
 * pixel's CombineReservoirBuffers instead of a launch of its own — in the exact mode, and in the fast mode of scenes without a dielectric / clear-coat / anisotropic material; 0: two
 * launches; identical image). */
int lumen_mi_set_tuning(lumen_mi_renderer*, const char* key, int value);

/* ---- tile sharding (new functionality: the reference is single-GPU) */
/* render only [x0,x1) x [y0,y1) of the image; RNG streams stay those of the full image.  The window stays set across resolution
 * changes (a frame whose window does not fit the image fails with LUMEN_MI_ERR_INVALID); (0,0,0,0) returns to the whole image. */
int lumen_mi_set_window(lumen_mi_renderer*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
/* Temporal history across tile seams.  After a frame, [x0,x1) x [y0,y1) (global pixels inside the window) of the reservoirs the NEXT
 * frame's temporal reuse reads as "previous" (ReSTIRKernels.cu:1015-1121) is packed to / unpacked from device memory, 80 bytes per
 * pixel, enqueued on the renderer's stream.  A rank exports the part of its tile that lies in a neighbour's halo and imports its own
 * halo ring from the owners, so that reuse across seams sees what a single GPU would (lumenrenderer_amd/tiles.py exchange_history). */
int lumen_mi_export_history(lumen_mi_renderer*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* device_dst);
int lumen_mi_import_history(lumen_mi_renderer*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, const void* device_src);
/* The reference swaps its reservoir buffers once per wave that holds a ray anywhere in the image (WaveFrontRenderer.cpp:697,827).  A
 * rank sees only its window: export writes the number of waves the last frame executed to a device int32, the caller takes the
 * maximum over the ranks (all-reduce), import advances the swap chain by the difference.  Call before lumen_mi_export_history. */
int lumen_mi_export_wave_count(lumen_mi_renderer*, void* device_i32);
int lumen_mi_import_wave_count(lumen_mi_renderer*, const void* device_i32);
/* The part of the render window this renderer OWNS (global pixel coordinates, inside the window; an empty rectangle = the whole
 * window).  Pixels of the window outside it are halo: they are rendered as far as the owned pixels' ReSTIR reuse needs them (surface
 * data, candidates, temporal pass, first reuse pass within 30 pixels), but get no indirect light, no second reuse pass and no
 * combine, and their radiance is undefined.  Owned pixels are unaffected (no reference equivalent: the reference is single-GPU). */
int lumen_mi_set_tile(lumen_mi_renderer*, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);

/* ---- tile groups: N processes, one per GPU, render ONE image (new functionality; north_star: "frames shard by tile across the 8 GPUs of one node with RCCL
 * gather over xGMI of the final radiance buffer").  csrc/group.cpp: the tile / window / halo plan, the seam exchange of temporal history, and the gather, in
 * C++ behind this ABI.  The plan functions are pure (no GPU).  Rectangles are x0 y0 x1 y1 in global pixels; an empty rectangle is all zero.
 *   plan:   a cols x rows grid (the one whose largest window is smallest); a rank owns `tile` and renders `window` = tile + 60-px halo clipped to the image
 *           (two spatial reuse passes x 30 px); every tile is sent in the common shape max_tile_w x max_tile_h.
 *   seams:  per peer, `send` = my tile inside the peer's window, `recv` = the peer's tile inside my window: what is exchanged after a frame at odd path depths. */
typedef struct lumen_mi_tile_plan { uint32_t cols, rows, halo; uint32_t tile[4]; uint32_t window[4]; uint32_t max_tile_w, max_tile_h; } lumen_mi_tile_plan;
typedef struct lumen_mi_seam { uint32_t peer; uint32_t send[4]; uint32_t recv[4]; } lumen_mi_seam;
int lumen_mi_group_plan(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, lumen_mi_tile_plan* out);
int lumen_mi_group_seams(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, lumen_mi_seam* out, uint32_t capacity, uint32_t* count);   /* out may be NULL: count only */
/* Transport.  Default: RCCL (two communicators: seam traffic on the renderer's stream, the gather on a stream of its own) — rank 0 obtains the id and hands the
 * bytes to the other ranks by any means (a file, an environment variable, MPI, a socket).  RCCL is resolved when first needed (dlopen "librccl.so.1", or the path
 * in LUMEN_MI_RCCL_LIBRARY): the library has no link-time dependency on it.  Alternative: a HOST transport — callbacks on host buffers; the group stages device data
 * through pinned memory around them.  exchange() posts all operations of one call together and returns when all have completed (ncclGroupStart / End semantics;
 * matching send / recv pairs appear in the same call on both sides); both callbacks return 0 on success. */
#define LUMEN_MI_GROUP_ID_BYTES 256
typedef struct lumen_mi_transport_op { uint32_t peer; int32_t send; void* host; size_t bytes; } lumen_mi_transport_op;      /* send != 0: to peer; else from peer */
typedef struct lumen_mi_transport {
    void* user;
    int (*exchange)(void* user, uint32_t n_ops, const lumen_mi_transport_op* ops);
    int (*allreduce_max_i32)(void* user, int32_t* value);
} lumen_mi_transport;
typedef struct lumen_mi_group lumen_mi_group;
int lumen_mi_group_unique_id(uint8_t id[LUMEN_MI_GROUP_ID_BYTES]);
/* The renderer must be initialised at the FULL image resolution; the group sets its window and tile.  id: the bytes of rank 0's lumen_mi_group_unique_id (may be NULL for
 * world == 1: then no communicator is created; with an id, world == 1 runs over RCCL too).  transport: NULL = RCCL.  Collective: every rank calls it. */
int lumen_mi_group_create(lumen_mi_renderer*, uint32_t rank, uint32_t world, const uint8_t* id, const lumen_mi_transport* transport, lumen_mi_group** out);
int lumen_mi_group_destroy(lumen_mi_group*);          /* the renderer returns to the whole image */
int lumen_mi_group_get_plan(lumen_mi_group*, lumen_mi_tile_plan* out);
/* one 1-element all-reduce per communicator and one full-size gather of empty tiles, synchronised: a rank that cannot reach its peers fails here, naming itself */
int lumen_mi_group_self_test(lumen_mi_group*, float* milliseconds);
/* TraceFrame of this rank's window (enqueue only); at odd path depths followed by the wave-count agreement and ONE grouped exchange of the halo rings' reservoirs,
 * all ordered on the renderer's stream ahead of the next frame */
int lumen_mi_group_trace_frame(lumen_mi_group*);
/* The rank's tile of the merged radiance -> one of two send tiles -> rank 0, on the gather stream (grouped send / recv), where each tile is placed in the assembled
 * frame; enqueue only.  Two send tiles and two assembled frames alternate, so the gather of frame i overlaps the rendering of frame i + 1. */
int lumen_mi_group_gather(lumen_mi_group*);
int lumen_mi_group_synchronize(lumen_mi_group*);
int lumen_mi_group_get_frame(lumen_mi_group*, float* rgba32f, size_t capacity_bytes);     /* rank 0: the last gathered frame, width x height RGBA32F (waits for it) */
int lumen_mi_group_frame_device(lumen_mi_group*, void** device_rgba32f);                  /* rank 0: the same, as a device pointer valid until the gather after next */
int lumen_mi_group_get_stats(lumen_mi_group*, uint64_t* gathers, float* mean_gather_ms); /* gather time = HIP events on the gather stream around transport + placement, sampled when a send tile is reused */

/* ---- ray-query seam (OptixWrapper::TraceRays, LumenPT/src/Framework/OptixWrapper.h:58-81): host arrays in, host arrays out */
/* (with lumen_mi_set_instrumented on, lumen_mi_query_closest leaves ITS traversal statistics — counters [20]..[46] — for lumen_mi_get_counters, in place of the last frame's) */
int lumen_mi_query_closest(lumen_mi_renderer*, uint32_t n, const float* origins3, const float* directions3, float tmin, float tmax,
                           uint32_t* instance_prim2, float* uvt3);
int lumen_mi_query_any(lumen_mi_renderer*, uint32_t n, const float* origins3, const float* directions3, float tmin, const float* tmax, uint8_t* occluded);

/* ---- known-answer hooks: run the device BSDF / math, device functions and whole kernels of the hot path on host arrays.  TEST SURFACE, not part of the reference's
 * interface: present in the default build (the test suite needs them); `make -C lumenrenderer_amd/csrc HOOKS=0` builds the library without any lumen_mi_test_* symbol,
 * without csrc/kat.cpp and without the hook kernels (csrc/lm_hooks.h). */
int lumen_mi_test_bsdf(lumen_mi_renderer*, uint32_t n, int mode, const float* mat23, const float* N, const float* T, const float* wo, const float* aux3, float* out8);
int lumen_mi_test_math(lumen_mi_renderer*, uint32_t n, int fn, const float* x, const float* y, float* out);
/* Known-answer hook for the device-side Reservoir::Update / UpdateWeight (ReSTIRData.h:115-163), CDF::Get (ReSTIRData.h:230-306) and
 * make_color (vendor/Include/Cuda/cuda/helpers.h:35-66).  mode 0: n sequences of 8 updates, a = weights, b = pdfs, c = seeds (8 n each),
 * out[33 n] = per update (weightSum, sampleCount, id held, taken), then the weight; mode 1: a = n prefix sums, b = m values, out[2 m] =
 * (index bits, pdf); mode 2: a = n linear values, out[n] = sRGB8 levels.
 * modes 3 / 5: Resample (LumenPT/src/CUDAKernels/ReSTIRKernels.cu:1259-1325) in the exact / fast arithmetic policy: a = n surfaces (35 floats:
 * position normal tangent incoming + the 23 material floats of lumen_mi_test_bsdf), b = n light samples (14: radiance normal position area
 * contribution solidAnglePdf), out[5 n] = (contribution, solidAnglePdf, applies).  modes 4 / 6: CombineBiased of two reservoirs
 * (ReSTIRKernels.cu:1200-1257), exact / fast: a = n surfaces, b = 2 n reservoirs (17: weightSum sampleCount weight sample(14)), c = n seeds,
 * out[18 n] = (reservoir(17), applies).  applies = 0 only in the fast modes, for surfaces the contracted evaluation does not cover (the
 * renderer scores those with the exact policy in a second launch).
 * lumen_mi_test_bsdf mode 2 = the contracted EvaluateBSDF of the fast policy (disney.cuh:320-405 for the isotropic opaque stack):
 * out8 = (bsdf, pdf, applies, 0, 0, 0). */
int lumen_mi_test_restir(lumen_mi_renderer*, int mode, uint32_t n, const float* a, const float* b, const uint32_t* c, uint32_t m, float* out);
/* Known-answer hooks that run whole KERNELS of the hot path on rows (csrc/kat.cpp; rows: tests/golden/ref_kat5.npz = what the reference's own __global__ kernel
 * bodies computed on a 64 x 48 synthetic image, generator oracle/ref_kat/gen_kat5.cpp).  Arrays are 32-bit words: floats by bit pattern, flags / counts / indices as
 * integers.  surface(40) = flags t position normal tangent incoming transport mat23; reservoir(17) = weightSum sampleCount weight radiance normal position area
 * contribution solidAnglePdf; light(16) = p0 p1 p2 normal radiance area (sorted by mean radiance), cdf = its prefix sums.
 * lumen_mi_test_restir_frame: one ReSTIR::Run (Framework/ReSTIR.cpp:65-233) — FillLightBags, PickPrimarySamples + GenerateShadowRay, temporal reuse, both spatial
 * passes, CombineReservoirBuffers (ReSTIRKernels.cu:343-370,402-522,546-582,787-980,1015-1121,1407-1436), launched as a frame launches them, on synthetic
 * surfaces; the visibility programs (closed, OptiX) are replaced by the masks occluded0 / occluded1 (per pixel, pass 1 / pass 2).  surf_prev40 NULL = the zero-filled
 * buffer of a first frame.  res4 [4][n][17] in / out: the four reservoir buffers.  bags [50000][2] (light index, pdf) or NULL.  stages [5][n][17]: the buffer each
 * kernel wrote right after it (pick, temporal, spatial 1, spatial 2, combine).  rays [2][n][8] (pixel, origin, direction, distance) in queue order, ray_counts[2].
 * direct [n][4]: the DIRECT channel (the three ShadeReservoirs passes, ReSTIRKernels.cu:600-665).  fast: 0 exact arithmetic, 2 the fast policy (both launches). */
int lumen_mi_test_restir_frame(lumen_mi_renderer*, uint32_t W, uint32_t H, const uint32_t* surf_cur40, const uint32_t* surf_prev40, const uint32_t* motion_half2,
                               uint32_t n_lights, const uint32_t* lights16, const uint32_t* cdf, uint32_t a_seed, int current_index, const uint8_t* occluded0,
                               const uint8_t* occluded1, int fast, uint32_t* res4, uint32_t* bags, uint32_t* stages, uint32_t* rays, uint32_t* ray_counts, uint32_t* direct);
/* ShadeDirect (GPUShadeDirect.cu:42-153) / ShadeIndirect (GPUShadeIndirect.cu:7-146) as the wave kernels call them, on rows (x, y, seed, surface(40)).
 * direct12 = emitted origin direction maxDistance radiance channel; indirect10 = emitted origin direction contribution; either may be NULL.  fast: tuning key fast_shade. */
int lumen_mi_test_shade(lumen_mi_renderer*, uint32_t n, uint32_t W, uint32_t H, const uint32_t* rows43, uint32_t n_lights, const uint32_t* lights16, const uint32_t* cdf,
                        int fast, uint32_t* direct12, uint32_t* indirect10);
/* ExtractSurfaceDataGpu (GPUExtractSurfaceData.cu:8-228) as every wave kernel runs it, on (hit record, ray) rows against the renderer's CURRENT scene (rows of
 * tests/golden/ref_kat6.npz: a scene of 1 x 1 textures, where no texture filtering can happen).  hits9 per row: table entry, primitive-local triangle, barycentric u, v
 * (binary16 bits), t, pixel x, y, two unused; rays9: origin direction contribution.  out35: flags t position normal geomNormal(0) tangent incoming transport color4 tint4
 * transmittance4 params3.  lumen_mi_test_extract0 runs the depth-0 KERNEL (surface extraction + GenerateMotionVector MotionVectors.cu:8-55 + ResolveDirectLightHits
 * GPUShadeDirect.cu:11-40, fused) on hit records for every pixel of the render resolution (row-major): G-buffer records [n][8][4], motion vectors (half2 bits), DIRECT [n][4].
 * lumen_mi_test_extract0 runs on the renderer's live frame buffers and INVALIDATES its history: the next TraceFrame starts as after a resize (reservoirs reset, frame counter 0). */
int lumen_mi_test_extract(lumen_mi_renderer*, uint32_t n, const uint32_t* hits9, const uint32_t* rays9, uint32_t* out35);
/* tex2D<float4>(texture object, u, v) as ExtractSurfaceDataGpu fetches it (GPUExtractSurfaceData.cu:59-60,169-181; texture object of PTTexture.cpp:35-74: linear filter,
 * wrap, normalised float read, sRGB decode per texel when created with normalize) on n coordinates uv2 of one texture -> out4 [n][4].  The filter is the rule the CUDA C
 * Programming Guide publishes for that unit (weights in 1.8 fixed point); tuning key "tex_filter" 1 selects unquantised fp32 weights instead. */
int lumen_mi_test_tex2d(lumen_mi_renderer*, lumen_mi_handle texture, uint32_t n, const float* uv2, float* out4);
int lumen_mi_test_extract0(lumen_mi_renderer*, const uint32_t* hits9, const uint32_t* dirs3, const uint32_t* eye3, const uint32_t* matrix16, float* gbuffer, uint32_t* motion, float* direct);
/* GeneratePrimaryRay (GPUGeneratePrimRay.cu:28-82): the primary-ray kernel on a W x H image; cam = U V W eye; out11 per pixel = x y origin direction contribution */
int lumen_mi_test_primary_rays(lumen_mi_renderer*, uint32_t W, uint32_t H, uint32_t frame_count, const uint32_t* cam_uvw_eye12, uint32_t* out11);
/* Known-answer hook for the host-side camera arithmetic of a frame (no renderer, no GPU): the image-plane vectors U, V, W of
 * Camera::GetVectorData (Lumen/src/Lumen/Renderer/Camera.cpp:79-93,122-128) for the rotation columns right / up / forward, and the matrix
 * projection * inverse(previous camera world matrix) the motion-vector pass receives (WaveFrontRenderer.cpp:763-776,
 * CPUShadingKernels.cu:39), both row major.  out25 = U(3) V(3) W(3) M(16). */
int lumen_mi_test_camera(const float right[3], const float up[3], const float forward[3], const float prev_world16[16], float fov_y_degrees, float aspect, float out25[25]);
/* host-side scene products, for tests: world-space triangles (9 floats each) and the sorted light list (16 floats each) + CDF */
int lumen_mi_get_world_triangles(lumen_mi_renderer*, float* out, uint32_t capacity_triangles, uint32_t* count);
int lumen_mi_get_lights(lumen_mi_renderer*, float* lights16, float* cdf, uint32_t capacity, uint32_t* count);
/* nodes / max_depth describe the tree as it was BUILT: binary nodes and binary depth of the host SAH build; 4-wide nodes and the number of wide levels for a tree
 * assembled from per-mesh trees or built on the device (tuning key gpu_build, which applies to full builds only: meshes added later are built per mesh on the host). */
int lumen_mi_get_bvh_info(lumen_mi_renderer*, uint32_t* nodes, uint32_t* triangles, uint32_t* max_depth);

#ifdef __cplusplus
}
#endif
#endif
