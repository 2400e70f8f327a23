"""ctypes binding of oracle/liblumen_oracle.so — TEST INFRASTRUCTURE ONLY (never imported by the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class MaterialDesc(C.Structure):
    _fields_ = [("diffuse_color", C.c_float * 4), ("emission", C.c_float * 3),
                ("tex_diffuse", C.c_int32), ("tex_normal", C.c_int32), ("tex_metal_rough", C.c_int32), ("tex_emissive", C.c_int32),
                ("tex_transmission", C.c_int32), ("tex_clearcoat", C.c_int32), ("tex_clearcoat_rough", C.c_int32), ("tex_tint", C.c_int32),
                ("transmission", C.c_float), ("clearcoat", C.c_float), ("clearcoat_roughness", C.c_float), ("ior", C.c_float),
                ("specular", C.c_float), ("specular_tint", C.c_float), ("subsurface", C.c_float), ("luminance", C.c_float),
                ("anisotropic", C.c_float), ("sheen", C.c_float), ("sheen_tint", C.c_float), ("metallic", C.c_float), ("roughness", C.c_float),
                ("tint", C.c_float * 3), ("transmittance", C.c_float * 3)]


def build():
    if os.environ.get("LUMEN_ORACLE_SO"):                    # e.g. a sanitizer build of the same sources (tests/test_cpu_host.py)
        return os.environ["LUMEN_ORACLE_SO"]
    so = os.path.join(ORACLE_DIR, "liblumen_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("lumen_oracle.cpp", "lumen_oracle.h", "orc_math.h", "orc_bsdf.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp, u32p, u8p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.orc_create.restype = C.c_void_p
        for name, args, res in [
            ("orc_destroy", [C.c_void_p], None), ("orc_set_threads", [C.c_void_p, C.c_int], None), ("orc_set_tex_filter", [C.c_void_p, C.c_int], None),
            ("orc_add_texture", [C.c_void_p, u8p, C.c_uint32, C.c_uint32, C.c_int], C.c_int),
            ("orc_kat_tex2d", [C.c_void_p, C.c_int, C.c_uint32, fp, fp], None),
            ("orc_add_material", [C.c_void_p, C.POINTER(MaterialDesc)], C.c_int),
            ("orc_add_primitive", [C.c_void_p, fp, C.c_uint32, u32p, C.c_uint32, C.c_int], C.c_int),
            ("orc_add_mesh", [C.c_void_p, i32p, C.c_uint32], C.c_int),
            ("orc_add_instance", [C.c_void_p, C.c_int, fp, C.c_int, fp, C.c_float, C.c_int], C.c_int),
            ("orc_set_instance_transform", [C.c_void_p, C.c_int, fp], None),
            ("orc_set_instance_emissiveness", [C.c_void_p, C.c_int, C.c_int, fp, C.c_float], None),
            ("orc_set_instance_override_material", [C.c_void_p, C.c_int, C.c_int], None),
            ("orc_get_denoiser_inputs", [C.c_void_p, C.c_float, C.c_float, fp, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)], None),
            ("orc_set_camera", [C.c_void_p, fp, fp, fp, fp, C.c_float], None),
            ("orc_set_resolution", [C.c_void_p, C.c_uint32, C.c_uint32], None),
            ("orc_set_depth", [C.c_void_p, C.c_uint32], None), ("orc_set_blend", [C.c_void_p, C.c_int], None),
            ("orc_set_window", [C.c_void_p] + [C.c_uint32] * 4, None),
            ("orc_trace_frame", [C.c_void_p], C.c_int),
            ("orc_get_radiance", [C.c_void_p, fp], None), ("orc_get_channel", [C.c_void_p, C.c_int, fp], None),
            ("orc_get_output_pixels", [C.c_void_p, u8p], None),
            ("orc_get_stats", [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32], None),
            ("orc_wang_hash", [C.c_uint32], C.c_uint32),
            ("orc_random_floats", [C.c_uint32, C.c_uint32, fp, u32p], None),
            ("orc_halton", [C.c_uint32, C.c_uint32], C.c_float),
            ("orc_pack_material", [fp, u32p, fp], None),
            ("orc_eval_bsdf", [C.c_uint32, fp, fp, fp, fp, fp, fp], None),
            ("orc_sample_bsdf", [C.c_uint32, fp, fp, fp, fp, fp, fp], None),
            ("orc_det_math", [C.c_uint32, C.c_int, fp, fp, fp], None),
            ("orc_f32_to_f16", [C.c_float], C.c_uint16), ("orc_f16_to_f32", [C.c_uint16], C.c_float),
            ("orc_camera_vectors", [fp, fp, fp, C.c_float, C.c_float, fp], None),
            ("orc_motion_matrix", [fp, C.c_float, C.c_float, fp], None),
            ("orc_reservoir_sequence", [C.c_uint32, fp, fp, u32p, fp, C.POINTER(C.c_int64), i32p, i32p, fp, fp], None),
            ("orc_cdf_get", [C.c_uint32, fp, C.c_uint32, fp, u32p, fp], None),
            ("orc_resample", [C.c_uint32, fp, fp, fp], None),
            ("orc_combine_biased", [C.c_uint32, C.c_uint32, fp, u32p, fp, fp], None),
            ("orc_combine_unbiased", [C.c_uint32, C.c_uint32, fp, u32p, fp, fp, fp], None),
            ("orc_make_color", [C.c_uint32, fp, u8p], None),
            ("orc_trace_closest", [C.c_void_p, C.c_uint32, fp, fp, C.c_float, C.c_float, u32p, fp, C.c_int], None),
            ("orc_trace_any", [C.c_void_p, C.c_uint32, fp, fp, C.c_float, fp, u8p, C.c_int], None),
            ("orc_world_triangles", [C.c_void_p, fp], C.c_uint32),
            ("orc_lights", [C.c_void_p, fp, fp], C.c_uint32),
            ("orc_get_gbuffer", [C.c_void_p, fp], None),
            ("orc_kat_extract", [C.c_void_p, C.c_uint32, u32p, u32p, u32p], None),
            ("orc_kat_motion_vectors", [C.c_uint32, C.c_uint32, u32p, u32p, u32p], None),
            ("orc_kat_resolve", [C.c_uint32, u32p, u32p, u32p], None),
            ("orc_kat_emissives", [C.c_void_p, C.c_int, u8p], C.c_uint32),
            ("orc_kat_light_slots", [C.c_void_p, u32p, C.c_uint32], C.c_uint32),
            ("orc_kat_light_weights", [C.c_uint32, u32p, u32p], None),
            ("orc_kat_primary_rays", [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p], None),
            ("orc_kat_shade", [C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p, u32p, u32p, u32p], None),
            ("orc_kat_restir_frame", [C.c_uint32, C.c_uint32, u32p, u32p, u32p, C.c_uint32, u32p, u32p, C.c_uint32, C.c_int, u8p, u8p, u32p, u32p, u32p, u32p, u32p, u32p, u32p], None),
        ]:
            f = getattr(L, name); f.argtypes = args; f.restype = res
        _LIB = L
    return _LIB


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u32ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can see 256 CPUs and be
    allowed the time of 16; more threads than that only add time-slicing)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                  # cgroup v2
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())   # v1
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


class Oracle:
    """Scene-level handle; mirrors the product's renderer facade closely enough that one scene
    description (lumenrenderer_amd.scenes.SceneDescription) can be replayed into either."""

    def __init__(self, threads=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        self.L.orc_set_threads(self.h, threads or usable_cpus())
        self.w = self.h_ = 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h); self.h = None

    def add_texture(self, rgba8, srgb):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        return self.L.orc_add_texture(self.h, u8ptr(a), a.shape[1], a.shape[0], int(bool(srgb)))

    def add_material(self, **kw):
        d = material_desc(**kw)
        return self.L.orc_add_material(self.h, C.byref(d))

    def add_primitive(self, vertices, indices, material):
        v = f32(vertices).reshape(-1, 12); i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        return self.L.orc_add_primitive(self.h, fptr(v), v.shape[0], u32ptr(i), i.size, material)

    def add_mesh(self, prims):
        p = np.ascontiguousarray(prims, dtype=np.int32)
        return self.L.orc_add_mesh(self.h, p.ctypes.data_as(C.POINTER(C.c_int32)), p.size)

    def add_instance(self, mesh, transform=None, emission_mode=0, override_radiance=(0, 0, 0), scale=1.0, override_material=-1):
        t = f32(np.eye(4) if transform is None else transform).reshape(16); r = f32(override_radiance)
        return self.L.orc_add_instance(self.h, mesh, fptr(t), emission_mode, fptr(r), float(scale), override_material)

    def set_instance_transform(self, inst, transform):
        t = f32(transform).reshape(16); self.L.orc_set_instance_transform(self.h, inst, fptr(t))

    def set_instance_emissiveness(self, inst, mode, override_radiance=(0, 0, 0), scale=1.0):
        self.L.orc_set_instance_emissiveness(self.h, inst, mode, fptr(f32(override_radiance)), float(scale))

    def set_instance_override_material(self, inst, material): self.L.orc_set_instance_override_material(self.h, inst, material)

    def set_camera(self, pos, right, up, forward, fov=90.0):
        self.L.orc_set_camera(self.h, fptr(f32(pos)), fptr(f32(right)), fptr(f32(up)), fptr(f32(forward)), float(fov))

    def set_resolution(self, w, h):
        self.w, self.h_ = w, h; self.L.orc_set_resolution(self.h, w, h)

    def set_depth(self, d): self.L.orc_set_depth(self.h, d)
    def set_blend(self, b): self.L.orc_set_blend(self.h, int(b))
    def set_window(self, x0, y0, x1, y1): self.L.orc_set_window(self.h, x0, y0, x1, y1)
    def set_tex_filter(self, mode): self.L.orc_set_tex_filter(self.h, int(mode))

    def tex2d(self, texture, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2); out = np.zeros((uv.shape[0], 4), np.float32)
        self.L.orc_kat_tex2d(self.h, texture, uv.shape[0], fptr(uv), fptr(out))
        return out
    def trace_frame(self): return self.L.orc_trace_frame(self.h)

    def radiance(self):
        out = np.zeros((self.h_, self.w, 4), np.float32); self.L.orc_get_radiance(self.h, fptr(out)); return out

    def channel(self, ch):
        out = np.zeros((self.h_, self.w, 4), np.float32); self.L.orc_get_channel(self.h, ch, fptr(out)); return out

    def output_pixels(self):
        out = np.zeros((self.h_, self.w, 4), np.uint8); self.L.orc_get_output_pixels(self.h, u8ptr(out)); return out

    def stats(self, n=16):
        out = (C.c_uint64 * n)(); self.L.orc_get_stats(self.h, out, n); return list(out)

    def denoiser_inputs(self, min_distance=0.1, max_distance=1000.0):
        n = self.h_ * self.w
        depth = np.zeros(n, np.float32); nr = np.zeros((n, 4), np.uint16); mv = np.zeros((n, 2), np.uint16)
        self.L.orc_get_denoiser_inputs(self.h, min_distance, max_distance, fptr(depth), nr.ctypes.data_as(C.POINTER(C.c_uint16)), mv.ctypes.data_as(C.POINTER(C.c_uint16)))
        return depth, nr, mv

    def gbuffer(self):
        out = np.zeros((self.h_, self.w, 8, 4), np.float32); self.L.orc_get_gbuffer(self.h, fptr(out)); return out

    def trace_closest(self, o, d, tmin=0.01, tmax=5000.0, use_bvh=True):
        o = f32(o).reshape(-1, 3); d = f32(d).reshape(-1, 3); n = o.shape[0]
        ip = np.zeros((n, 2), np.uint32); uvt = np.zeros((n, 3), np.float32)
        self.L.orc_trace_closest(self.h, n, fptr(o), fptr(d), tmin, tmax, u32ptr(ip), fptr(uvt), int(use_bvh))
        return ip, uvt

    def trace_any(self, o, d, tmax, tmin=0.01, use_bvh=True):
        o = f32(o).reshape(-1, 3); d = f32(d).reshape(-1, 3); tm = f32(tmax).ravel(); n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        self.L.orc_trace_any(self.h, n, fptr(o), fptr(d), tmin, fptr(tm), u8ptr(occ), int(use_bvh))
        return occ

    def world_triangles(self):
        n = self.L.orc_world_triangles(self.h, None); out = np.zeros((n, 3, 3), np.float32)
        self.L.orc_world_triangles(self.h, fptr(out)); return out

    def lights(self):
        n = self.L.orc_lights(self.h, None, None); out = np.zeros((n, 16), np.float32); cdf = np.zeros(n, np.float32)
        self.L.orc_lights(self.h, fptr(out), fptr(cdf)); return out, cdf


def material_desc(diffuse_color=(1, 1, 1, 1), emission=(0, 0, 0), tex_diffuse=-1, tex_normal=-1, tex_metal_rough=-1, tex_emissive=-1,
                  tex_transmission=-1, tex_clearcoat=-1, tex_clearcoat_rough=-1, tex_tint=-1, transmission=0.0, clearcoat=0.0,
                  clearcoat_roughness=0.0, ior=1.0, specular=0.0, specular_tint=0.0, subsurface=0.0, luminance=1.0, anisotropic=0.0,
                  sheen=0.0, sheen_tint=0.0, metallic=1.0, roughness=1.0, tint=(1, 1, 1), transmittance=(1, 1, 1)):
    """Defaults = LumenRenderer::MaterialData() (Lumen/src/Lumen/Renderer/LumenRenderer.h:66-84)."""
    d = MaterialDesc()
    d.diffuse_color = (C.c_float * 4)(*diffuse_color); d.emission = (C.c_float * 3)(*emission)
    d.tex_diffuse, d.tex_normal, d.tex_metal_rough, d.tex_emissive = tex_diffuse, tex_normal, tex_metal_rough, tex_emissive
    d.tex_transmission, d.tex_clearcoat, d.tex_clearcoat_rough, d.tex_tint = tex_transmission, tex_clearcoat, tex_clearcoat_rough, tex_tint
    d.transmission, d.clearcoat, d.clearcoat_roughness, d.ior = transmission, clearcoat, clearcoat_roughness, ior
    d.specular, d.specular_tint, d.subsurface, d.luminance = specular, specular_tint, subsurface, luminance
    d.anisotropic, d.sheen, d.sheen_tint, d.metallic, d.roughness = anisotropic, sheen, sheen_tint, metallic, roughness
    d.tint = (C.c_float * 3)(*tint); d.transmittance = (C.c_float * 3)(*transmittance)
    return d
