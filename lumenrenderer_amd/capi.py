"""ctypes binding of include/lumen_mi.h (the drop-in C ABI).  No torch types cross this boundary."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, ERR_INVALID, ERR_DEVICE, ERR_STATE, NO_LIGHTS = 0, 1, 2, 3, 4


class LumenMIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lumen_mi error {code}: {msg}")
        self.code = code


class Settings(C.Structure):
    _fields_ = [("depth", C.c_uint32), ("render_width", C.c_uint32), ("render_height", C.c_uint32),
                ("output_width", C.c_uint32), ("output_height", C.c_uint32), ("blend_output", C.c_int32), ("device", C.c_int32)]


class MaterialData(C.Structure):
    _fields_ = [("diffuse_color", C.c_float * 4), ("emission", C.c_float * 3),
                ("diffuse_texture", C.c_uint64), ("normal_map", C.c_uint64), ("metallic_roughness_texture", C.c_uint64), ("emissive_texture", C.c_uint64),
                ("transmission_texture", C.c_uint64), ("clearcoat_texture", C.c_uint64), ("clearcoat_roughness_texture", C.c_uint64), ("tint_texture", C.c_uint64),
                ("transmission_factor", C.c_float), ("clearcoat_factor", C.c_float), ("clearcoat_roughness_factor", C.c_float), ("index_of_refraction", C.c_float),
                ("specular_factor", C.c_float), ("specular_tint_factor", C.c_float), ("subsurface_factor", C.c_float), ("luminance", C.c_float), ("anisotropic", C.c_float),
                ("sheen_factor", C.c_float), ("sheen_tint_factor", C.c_float), ("metallic_factor", C.c_float), ("roughness_factor", C.c_float),
                ("tint_factor", C.c_float * 3), ("transmittance", C.c_float * 3)]


class PrimitiveData(C.Structure):
    _fields_ = [("interleaved", C.c_int32), ("vertex_binary", C.c_void_p), ("positions", C.POINTER(C.c_float)), ("tex_coords", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("tangents", C.POINTER(C.c_float)), ("n_vertices", C.c_uint32), ("index_binary", C.c_void_p),
                ("n_indices", C.c_uint32), ("index_size", C.c_uint32), ("material", C.c_uint64)]


class TilePlan(C.Structure):
    _fields_ = [("cols", C.c_uint32), ("rows", C.c_uint32), ("halo", C.c_uint32), ("tile", C.c_uint32 * 4), ("window", C.c_uint32 * 4),
                ("max_tile_w", C.c_uint32), ("max_tile_h", C.c_uint32)]


class Seam(C.Structure):
    _fields_ = [("peer", C.c_uint32), ("send", C.c_uint32 * 4), ("recv", C.c_uint32 * 4)]


class TransportOp(C.Structure):
    _fields_ = [("peer", C.c_uint32), ("send", C.c_int32), ("host", C.c_void_p), ("bytes", C.c_size_t)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(TransportOp))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32))


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_max_i32", ALLREDUCE_FN)]


GROUP_ID_BYTES = 256


def library_path():
    """The in-tree library; LUMEN_MI_LIBRARY names another BUILD of the same sources (tools/ab_lib.sh: compile-time variants built beforehand and compared on one GPU box
    without rebuilding there).  Either way it is the HIP library: there is nothing else to load."""
    return os.environ.get("LUMEN_MI_LIBRARY") or os.path.join(_HERE, "liblumen_mi.so")


# every symbol include/lumen_mi.h declares: (name, argtypes); all return int except last_error
_FP, _U8P, _U32P, _U64P = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_R, _H = C.c_void_p, C.c_uint64
SYMBOLS = {
    "lumen_mi_create": [C.POINTER(C.c_void_p)], "lumen_mi_init": [_R, C.POINTER(Settings)], "lumen_mi_destroy": [_R],
    "lumen_mi_set_stream": [_R, C.c_void_p],
    "lumen_mi_create_texture": [_R, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, _U64P],
    "lumen_mi_create_material": [_R, C.POINTER(MaterialData), _U64P],
    "lumen_mi_update_material": [_R, _H, C.POINTER(MaterialData)],
    "lumen_mi_create_default_resources": [_R, _U64P, _U64P, _U64P],
    "lumen_mi_create_primitive": [_R, C.POINTER(PrimitiveData), _U64P, _U32P],
    "lumen_mi_create_mesh": [_R, _U64P, C.c_uint32, _U64P], "lumen_mi_create_scene": [_R, _U64P], "lumen_mi_set_scene": [_R, _H],
    "lumen_mi_scene_add_mesh": [_R, _H, _H, _U64P], "lumen_mi_scene_clear": [_R, _H],
    "lumen_mi_instance_set_transform": [_R, _H, _FP], "lumen_mi_instance_set_emissiveness": [_R, _H, C.c_int, _FP, C.c_float],
    "lumen_mi_instance_set_override_material": [_R, _H, _H],
    "lumen_mi_camera_set": [_R, _FP, _FP, _FP, _FP, C.c_float],
    "lumen_mi_set_render_resolution": [_R, C.c_uint32, C.c_uint32], "lumen_mi_set_output_resolution": [_R, C.c_uint32, C.c_uint32],
    "lumen_mi_get_render_resolution": [_R, _U32P, _U32P], "lumen_mi_get_output_resolution": [_R, _U32P, _U32P],
    "lumen_mi_set_blend_mode": [_R, C.c_int], "lumen_mi_get_blend_mode": [_R, C.POINTER(C.c_int)], "lumen_mi_set_depth": [_R, C.c_uint32],
    "lumen_mi_trace_frame": [_R], "lumen_mi_trace_frame_async": [_R], "lumen_mi_synchronize": [_R],
    "lumen_mi_start_rendering": [_R], "lumen_mi_stop_rendering": [_R], "lumen_mi_perform_deferred_operations": [_R],
    "lumen_mi_get_output_pixels": [_R, _U8P, C.c_size_t, _U32P, _U32P], "lumen_mi_get_radiance": [_R, _FP, C.c_size_t],
    "lumen_mi_get_radiance_half4": [_R, C.POINTER(C.c_uint16), C.c_size_t],
    "lumen_mi_copy_radiance_device": [_R, C.c_void_p], "lumen_mi_get_channel": [_R, C.c_int, _FP, C.c_size_t],
    "lumen_mi_copy_radiance_rect_device": [_R, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32],
    "lumen_mi_copy_rect_device": [_R, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32],
    "lumen_mi_get_gbuffer": [_R, _FP, C.c_size_t],
    "lumen_mi_get_frame_stat": [_R, C.c_char_p, _U64P], "lumen_mi_get_counters": [_R, _U64P, C.c_uint32],
    "lumen_mi_get_counter_totals": [_R, _U64P, C.c_uint32, C.c_int],
    "lumen_mi_get_kernel_time": [_R, C.c_int, _FP, _U32P], "lumen_mi_enable_kernel_timing": [_R, C.c_int], "lumen_mi_set_instrumented": [_R, C.c_int],
    "lumen_mi_set_tuning": [_R, C.c_char_p, C.c_int],
    "lumen_mi_get_denoiser_inputs": [_R, C.c_float, C.c_float, _FP, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)],
    "lumen_mi_set_window": [_R, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32],
    "lumen_mi_set_tile": [_R, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32],
    "lumen_mi_export_history": [_R, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p],
    "lumen_mi_import_history": [_R, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p],
    "lumen_mi_export_wave_count": [_R, C.c_void_p], "lumen_mi_import_wave_count": [_R, C.c_void_p],
    "lumen_mi_query_closest": [_R, C.c_uint32, _FP, _FP, C.c_float, C.c_float, _U32P, _FP],
    "lumen_mi_query_any": [_R, C.c_uint32, _FP, _FP, C.c_float, _FP, _U8P],
    "lumen_mi_test_bsdf": [_R, C.c_uint32, C.c_int, _FP, _FP, _FP, _FP, _FP, _FP], "lumen_mi_test_math": [_R, C.c_uint32, C.c_int, _FP, _FP, _FP],
    "lumen_mi_test_restir": [_R, C.c_int, C.c_uint32, _FP, _FP, _U32P, C.c_uint32, _FP],
    "lumen_mi_test_restir_frame": [_R, C.c_uint32, C.c_uint32, _U32P, _U32P, _U32P, C.c_uint32, _U32P, _U32P, C.c_uint32, C.c_int, _U8P, _U8P, C.c_int,
                                   _U32P, _U32P, _U32P, _U32P, _U32P, _U32P],
    "lumen_mi_test_shade": [_R, C.c_uint32, C.c_uint32, C.c_uint32, _U32P, C.c_uint32, _U32P, _U32P, C.c_int, _U32P, _U32P],
    "lumen_mi_test_extract": [_R, C.c_uint32, _U32P, _U32P, _U32P],
    "lumen_mi_test_tex2d": [_R, C.c_uint64, C.c_uint32, _FP, _FP],
    "lumen_mi_test_extract0": [_R, _U32P, _U32P, _U32P, _U32P, _FP, _U32P, _FP],
    "lumen_mi_test_primary_rays": [_R, C.c_uint32, C.c_uint32, C.c_uint32, _U32P, _U32P],
    "lumen_mi_test_camera": [_FP, _FP, _FP, _FP, C.c_float, C.c_float, _FP],
    "lumen_mi_get_world_triangles": [_R, _FP, C.c_uint32, _U32P], "lumen_mi_get_lights": [_R, _FP, _FP, C.c_uint32, _U32P],
    "lumen_mi_get_bvh_info": [_R, _U32P, _U32P, _U32P],
    "lumen_mi_group_plan": [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(TilePlan)],
    "lumen_mi_group_seams": [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Seam), C.c_uint32, _U32P],
    "lumen_mi_group_unique_id": [_U8P],
    "lumen_mi_group_create": [_R, C.c_uint32, C.c_uint32, _U8P, C.POINTER(Transport), C.POINTER(C.c_void_p)],
    "lumen_mi_group_destroy": [C.c_void_p], "lumen_mi_group_get_plan": [C.c_void_p, C.POINTER(TilePlan)],
    "lumen_mi_group_self_test": [C.c_void_p, _FP],
    "lumen_mi_group_trace_frame": [C.c_void_p], "lumen_mi_group_gather": [C.c_void_p], "lumen_mi_group_synchronize": [C.c_void_p],
    "lumen_mi_group_get_frame": [C.c_void_p, _FP, C.c_size_t], "lumen_mi_group_frame_device": [C.c_void_p, C.POINTER(C.c_void_p)],
    "lumen_mi_group_get_stats": [C.c_void_p, _U64P, _FP],
}


def _share_the_hip_runtime_with_torch():
    """PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and load them by a path-qualified name.  If
    liblumen_mi.so (NEEDED libamdhip64.so.7, found under /opt/rocm/lib) is loaded BEFORE torch, the process ends up with two HIP
    runtimes and the second one finds no GPU ("No HIP GPUs are available"), and device pointers could not be shared either.
    Loading torch's copy first (same SONAME) makes both bind to one runtime whichever is imported first.  Without a torch install
    nothing happens and the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                   # fall back to the system runtime; torch, if imported later, reports the clash itself


def load_library():
    """Load liblumen_mi.so.  Fails loudly when it has not been built: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LumenMIError(ERR_STATE, f"{path} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(make -C lumenrenderer_amd/csrc); the HIP library is the only implementation")
    # four busy HIP streams + whatever the host adds (RCCL): HIP's default of 4 hardware queues makes busy streams share one and
    # serialise; takes effect only if HIP has not been initialised in this process yet
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _share_the_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, args in SYMBOLS.items():
        f = getattr(lib, name, None)
        if f is None:                                  # an older build selected with LUMEN_MI_LIBRARY (A/B runs); __graft_entry__.build() and the CPU suite hold the
            continue                                   # in-tree library to the full symbol list of include/lumen_mi.h
        f.argtypes = args
        f.restype = C.c_int
    lib.lumen_mi_last_error.argtypes = []
    lib.lumen_mi_last_error.restype = C.c_char_p
    _LIB = lib
    return lib


def check(lib, rc, allow=()):
    if rc != OK and rc not in allow:
        raise LumenMIError(rc, (lib.lumen_mi_last_error() or b"").decode("utf-8", "replace"))
    return rc
