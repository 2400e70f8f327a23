"""Host-side mirror of the reference's renderer interface over the C ABI.

``LumenRendererMI`` keeps the method names, argument meaning and call order of ``LumenRenderer`` /
``WaveFront::WaveFrontRenderer`` (reference: Lumen/src/Lumen/Renderer/LumenRenderer.h:37-219,
LumenPT/src/Framework/WaveFrontRenderer.h:31-269) so that a Sandbox-style driver reads the same:
``Init -> CreateDefaultResources -> CreateTexture/CreateMaterial/CreatePrimitive/CreateMesh/CreateScene ->
scene.AddMesh -> TraceFrame (or StartRendering) -> GetOutputTexturePixels``.  Errors surface as ``LumenMIError``
(the reference asserts/aborts instead, CudaUtilities.h:24-28).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import MaterialData, PrimitiveData, Settings, check

EMISSION_ENABLED, EMISSION_DISABLED, EMISSION_OVERRIDE = 0, 1, 2


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class MeshInstance:
    """Lumen::MeshInstance (Lumen/src/Lumen/ModelLoading/MeshInstance.h:22-112)."""

    def __init__(self, renderer, handle):
        self._r, self.handle = renderer, handle

    def SetTransform(self, world_matrix_row_major):
        m = _f32(world_matrix_row_major).reshape(16)
        check(self._r.lib, self._r.lib.lumen_mi_instance_set_transform(self._r.h, self.handle, _fp(m)))

    def SetEmissiveness(self, mode, override_radiance=(0.0, 0.0, 0.0), scale=1.0):
        rad = _f32(override_radiance)
        check(self._r.lib, self._r.lib.lumen_mi_instance_set_emissiveness(self._r.h, self.handle, int(mode), _fp(rad), float(scale)))

    def SetOverrideMaterial(self, material):
        check(self._r.lib, self._r.lib.lumen_mi_instance_set_override_material(self._r.h, self.handle, material))


class Scene:
    """ILumenScene (Lumen/src/Lumen/ModelLoading/ILumenScene.h:48-67)."""

    def __init__(self, renderer, handle):
        self._r, self.handle = renderer, handle
        self.m_MeshInstances = []

    def AddMesh(self, mesh):
        out = C.c_uint64()
        check(self._r.lib, self._r.lib.lumen_mi_scene_add_mesh(self._r.h, self.handle, mesh, C.byref(out)))
        inst = MeshInstance(self._r, out.value)
        self.m_MeshInstances.append(inst)
        return inst

    def Clear(self):
        check(self._r.lib, self._r.lib.lumen_mi_scene_clear(self._r.h, self.handle))
        self.m_MeshInstances = []


class LumenRendererMI:
    def __init__(self):
        self.lib = capi.load_library()
        h = C.c_void_p()
        check(self.lib, self.lib.lumen_mi_create(C.byref(h)))
        self.h = h
        self.m_Scene = None
        self._defaults = None

    # ---- lifetime -------------------------------------------------------------------------------------------------
    def Init(self, depth=5, render_resolution=(1280, 720), output_resolution=None, blend_output=False, device=0):
        """WaveFrontRenderer::Init(const WaveFrontSettings&) — Sandbox defaults: Application.cpp:84-95."""
        out = output_resolution or render_resolution
        s = Settings(depth, render_resolution[0], render_resolution[1], out[0], out[1], int(blend_output), device)
        check(self.lib, self.lib.lumen_mi_init(self.h, C.byref(s)))

    def close(self):
        if self.h:
            self.lib.lumen_mi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        check(self.lib, self.lib.lumen_mi_set_stream(self.h, C.c_void_p(int(hip_stream_ptr))))

    # ---- resource factories ----------------------------------------------------------------------------------------
    def CreateTexture(self, pixel_data_rgba8, width=None, height=None, normalize=False):
        a = np.ascontiguousarray(pixel_data_rgba8, dtype=np.uint8)
        if width is None:
            height, width = a.shape[0], a.shape[1]
        out = C.c_uint64()
        check(self.lib, self.lib.lumen_mi_create_texture(self.h, a.ctypes.data_as(C.c_void_p), width, height, int(bool(normalize)), C.byref(out)))
        return out.value

    def CreateDefaultResources(self):
        w, n, d = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self.lib, self.lib.lumen_mi_create_default_resources(self.h, C.byref(w), C.byref(n), C.byref(d)))
        self._defaults = (w.value, n.value, d.value)
        return self._defaults

    def CreateMaterial(self, **kw):
        """Keyword names = fields of lumen_mi_material_data; defaults = LumenRenderer::MaterialData() (LumenRenderer.h:66-84)."""
        d = MaterialData()
        d.diffuse_color = (C.c_float * 4)(*kw.get("diffuse_color", (1, 1, 1, 1)))
        d.emission = (C.c_float * 3)(*kw.get("emission", (0, 0, 0)))
        for f in ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
                  "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture"):
            setattr(d, f, int(kw.get(f, 0)))
        defaults = dict(transmission_factor=0.0, clearcoat_factor=0.0, clearcoat_roughness_factor=0.0, index_of_refraction=1.0, specular_factor=0.0,
                        specular_tint_factor=0.0, subsurface_factor=0.0, luminance=1.0, anisotropic=0.0, sheen_factor=0.0, sheen_tint_factor=0.0,
                        metallic_factor=1.0, roughness_factor=1.0)
        for f, v in defaults.items():
            setattr(d, f, float(kw.get(f, v)))
        d.tint_factor = (C.c_float * 3)(*kw.get("tint_factor", (1, 1, 1)))
        d.transmittance = (C.c_float * 3)(*kw.get("transmittance", (1, 1, 1)))
        out = C.c_uint64()
        check(self.lib, self.lib.lumen_mi_create_material(self.h, C.byref(d), C.byref(out)))
        return out.value

    def CreatePrimitive(self, material, indices, vertices=None, positions=None, tex_coords=None, normals=None, tangents=None, index_size=4):
        """PrimitiveData: either interleaved 48-byte ``vertices`` (n x 12 floats) or separate attribute arrays."""
        d = PrimitiveData()
        keep = []
        if vertices is not None:
            v = _f32(vertices).reshape(-1, 12); keep.append(v)
            d.interleaved, d.vertex_binary, d.n_vertices = 1, v.ctypes.data_as(C.c_void_p), v.shape[0]
        else:
            p = _f32(positions).reshape(-1, 3); keep.append(p)
            d.interleaved, d.positions, d.n_vertices = 0, _fp(p), p.shape[0]
            for name, arr, width in (("tex_coords", tex_coords, 2), ("normals", normals, 3), ("tangents", tangents, 4)):
                if arr is not None:
                    a = _f32(arr).reshape(-1, width); keep.append(a); setattr(d, name, _fp(a))
        idx = np.ascontiguousarray(indices, dtype=np.uint16 if index_size == 2 else np.uint32).ravel(); keep.append(idx)
        d.index_binary, d.n_indices, d.index_size, d.material = idx.ctypes.data_as(C.c_void_p), idx.size, index_size, material
        out, nl = C.c_uint64(), C.c_uint32()
        check(self.lib, self.lib.lumen_mi_create_primitive(self.h, C.byref(d), C.byref(out), C.byref(nl)))
        return out.value, nl.value

    def CreateMesh(self, primitives):
        arr = (C.c_uint64 * len(primitives))(*primitives)
        out = C.c_uint64()
        check(self.lib, self.lib.lumen_mi_create_mesh(self.h, arr, len(primitives), C.byref(out)))
        return out.value

    def CreateScene(self):
        out = C.c_uint64()
        check(self.lib, self.lib.lumen_mi_create_scene(self.h, C.byref(out)))
        return Scene(self, out.value)

    def SetScene(self, scene):
        """``renderer->m_Scene = scene`` (Sandbox/src/Application.cpp:144)."""
        check(self.lib, self.lib.lumen_mi_set_scene(self.h, scene.handle))
        self.m_Scene = scene

    def SetCamera(self, position, right, up, forward, fov_y_degrees=90.0):
        p, r, u, f = _f32(position), _f32(right), _f32(up), _f32(forward)
        check(self.lib, self.lib.lumen_mi_camera_set(self.h, _fp(p), _fp(r), _fp(u), _fp(f), float(fov_y_degrees)))

    # ---- settings --------------------------------------------------------------------------------------------------
    def SetRenderResolution(self, w, h): check(self.lib, self.lib.lumen_mi_set_render_resolution(self.h, w, h))
    def SetOutputResolution(self, w, h): check(self.lib, self.lib.lumen_mi_set_output_resolution(self.h, w, h))
    def SetBlendMode(self, blend): check(self.lib, self.lib.lumen_mi_set_blend_mode(self.h, int(bool(blend))))
    def SetDepth(self, depth): check(self.lib, self.lib.lumen_mi_set_depth(self.h, depth))

    def GetRenderResolution(self):
        w, h = C.c_uint32(), C.c_uint32(); check(self.lib, self.lib.lumen_mi_get_render_resolution(self.h, C.byref(w), C.byref(h))); return w.value, h.value

    def GetOutputResolution(self):
        w, h = C.c_uint32(), C.c_uint32(); check(self.lib, self.lib.lumen_mi_get_output_resolution(self.h, C.byref(w), C.byref(h))); return w.value, h.value

    def GetBlendMode(self):
        b = C.c_int(); check(self.lib, self.lib.lumen_mi_get_blend_mode(self.h, C.byref(b))); return bool(b.value)

    def SetWindow(self, x0, y0, x1, y1): check(self.lib, self.lib.lumen_mi_set_window(self.h, x0, y0, x1, y1))
    def SetTile(self, x0, y0, x1, y1): check(self.lib, self.lib.lumen_mi_set_tile(self.h, x0, y0, x1, y1))

    def ExportHistory(self, rect, device_ptr):
        """Pack rect = (x0, y0, x1, y1) of the reservoirs the next frame reads as "previous" into device memory (80 bytes / pixel)."""
        check(self.lib, self.lib.lumen_mi_export_history(self.h, *[int(v) for v in rect], C.c_void_p(int(device_ptr))))

    def ExportWaveCount(self, device_ptr): check(self.lib, self.lib.lumen_mi_export_wave_count(self.h, C.c_void_p(int(device_ptr))))
    def ImportWaveCount(self, device_ptr): check(self.lib, self.lib.lumen_mi_import_wave_count(self.h, C.c_void_p(int(device_ptr))))

    def ImportHistory(self, rect, device_ptr):
        check(self.lib, self.lib.lumen_mi_import_history(self.h, *[int(v) for v in rect], C.c_void_p(int(device_ptr))))

    # ---- rendering -------------------------------------------------------------------------------------------------
    def TraceFrame(self):
        """Blocking frame; returns False when the frame was skipped because the scene has no lights."""
        return check(self.lib, self.lib.lumen_mi_trace_frame(self.h), allow=(capi.NO_LIGHTS,)) == capi.OK

    def TraceFrameAsync(self):
        return check(self.lib, self.lib.lumen_mi_trace_frame_async(self.h), allow=(capi.NO_LIGHTS,)) == capi.OK

    def Synchronize(self): check(self.lib, self.lib.lumen_mi_synchronize(self.h))
    def StartRendering(self): check(self.lib, self.lib.lumen_mi_start_rendering(self.h))
    def StopRendering(self): check(self.lib, self.lib.lumen_mi_stop_rendering(self.h))
    def PerformDeferredOperations(self): check(self.lib, self.lib.lumen_mi_perform_deferred_operations(self.h))

    # ---- readback --------------------------------------------------------------------------------------------------
    def _window_shape(self):
        # output of the last frame covers the render window
        w, h = C.c_uint32(), C.c_uint32()
        buf = (C.c_uint8 * 4)()
        self.lib.lumen_mi_get_output_pixels(self.h, buf, 0, C.byref(w), C.byref(h))
        if w.value == 0 or h.value == 0:            # (with a render thread running, the first frame may land between this query and the read-back)
            from .capi import LumenMIError
            raise LumenMIError(3, "no frame has been traced yet")
        return h.value, w.value

    def GetOutputTexturePixels(self):
        hh, ww = self._window_shape()
        out = np.zeros((hh, ww, 4), np.uint8)
        w, h = C.c_uint32(), C.c_uint32()
        check(self.lib, self.lib.lumen_mi_get_output_pixels(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.nbytes, C.byref(w), C.byref(h)))
        return out

    def MakeScreenshot(self, path, gamma=2.2):
        """Sandbox OutputLayer::MakeScreenshot (OutputLayer.cpp:882-896): gamma-corrected output of the last frame as a PNG file."""
        from . import screenshot
        return screenshot.make_screenshot(self, path, gamma)

    def GetRadiance(self):
        hh, ww = self._window_shape()
        out = np.zeros((hh, ww, 4), np.float32)
        check(self.lib, self.lib.lumen_mi_get_radiance(self.h, _fp(out), out.nbytes))
        return out

    def GetRadianceHalf4(self):
        """The merged radiance rounded once to binary16 — what the reference's half4 pixel buffers would hold (as float16 array)."""
        hh, ww = self._window_shape()
        out = np.zeros((hh, ww, 4), np.uint16)
        check(self.lib, self.lib.lumen_mi_get_radiance_half4(self.h, out.ctypes.data_as(C.POINTER(C.c_uint16)), out.nbytes))
        return out.view(np.float16)

    def GetChannel(self, ch):
        hh, ww = self._window_shape()
        out = np.zeros((hh, ww, 4), np.float32)
        check(self.lib, self.lib.lumen_mi_get_channel(self.h, ch, _fp(out), out.nbytes))
        return out

    def GetGBuffer(self):
        hh, ww = self._window_shape()
        out = np.zeros((hh, ww, 8, 4), np.float32)
        check(self.lib, self.lib.lumen_mi_get_gbuffer(self.h, _fp(out), out.nbytes))
        return out

    def CopyRadianceToDevice(self, device_ptr):
        check(self.lib, self.lib.lumen_mi_copy_radiance_device(self.h, C.c_void_p(int(device_ptr))))

    def CopyRadianceRectToDevice(self, rect, device_ptr, pitch):
        """The image rectangle (x0, y0, x1, y1) — inside the render window — of the merged radiance into a device image of `pitch` RGBA32F pixels per row, on the renderer's stream."""
        check(self.lib, self.lib.lumen_mi_copy_radiance_rect_device(self.h, int(rect[0]), int(rect[1]), int(rect[2]), int(rect[3]), C.c_void_p(int(device_ptr)), int(pitch)))

    def CopyRectDevice(self, dst_ptr, dst_pitch, src_ptr, src_pitch, w, h):
        """A w x h RGBA32F rectangle between two pitched device images on the renderer's stream (tile assembly)."""
        check(self.lib, self.lib.lumen_mi_copy_rect_device(self.h, C.c_void_p(int(dst_ptr)), int(dst_pitch), C.c_void_p(int(src_ptr)), int(src_pitch), int(w), int(h)))

    def GetCounters(self, n=24):
        out = (C.c_uint64 * n)(); check(self.lib, self.lib.lumen_mi_get_counters(self.h, out, n)); return list(out)

    def GetCounterTotals(self, n=50, reset=False):
        """Counters summed over every TraceFrame since creation / the last reset (layout of GetCounters; [3] = TraceFrames summed)."""
        out = (C.c_uint64 * n)(); check(self.lib, self.lib.lumen_mi_get_counter_totals(self.h, out, n, int(reset))); return list(out)

    def GetDenoiserInputs(self, min_distance=0.1, max_distance=1000.0):
        """(depth [h,w] f32, normal_roughness [h,w,4] f16 bits, motion [h,w,2] f16 bits) of the last frame's window."""
        h, w = self._window_shape(); n = h * w
        depth = np.zeros(n, np.float32); nr = np.zeros((n, 4), np.uint16); mv = np.zeros((n, 2), np.uint16)
        check(self.lib, self.lib.lumen_mi_get_denoiser_inputs(self.h, min_distance, max_distance, _fp(depth),
                                                              nr.ctypes.data_as(C.POINTER(C.c_uint16)), mv.ctypes.data_as(C.POINTER(C.c_uint16))))
        return depth.reshape(h, w), nr.reshape(h, w, 4), mv.reshape(h, w, 2)

    def GetLastFrameStat(self, key):
        v = C.c_uint64(); check(self.lib, self.lib.lumen_mi_get_frame_stat(self.h, key.encode(), C.byref(v))); return v.value

    def EnableKernelTiming(self, on=True): check(self.lib, self.lib.lumen_mi_enable_kernel_timing(self.h, int(on)))
    def SetInstrumented(self, on=True): check(self.lib, self.lib.lumen_mi_set_instrumented(self.h, int(on)))
    def SetTuning(self, key, value): check(self.lib, self.lib.lumen_mi_set_tuning(self.h, key.encode(), int(value)))

    def GetKernelTime(self, which):
        ms, n = C.c_float(), C.c_uint32(); check(self.lib, self.lib.lumen_mi_get_kernel_time(self.h, which, C.byref(ms), C.byref(n))); return ms.value, n.value

    # ---- ray-query seam / test hooks -----------------------------------------------------------------------------------
    def QueryClosest(self, origins, directions, tmin=0.01, tmax=5000.0):
        o, d = _f32(origins).reshape(-1, 3), _f32(directions).reshape(-1, 3); n = o.shape[0]
        ip, uvt = np.zeros((n, 2), np.uint32), np.zeros((n, 3), np.float32)
        check(self.lib, self.lib.lumen_mi_query_closest(self.h, n, _fp(o), _fp(d), tmin, tmax, ip.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(uvt)))
        return ip, uvt

    def QueryAny(self, origins, directions, tmax, tmin=0.01):
        o, d, tm = _f32(origins).reshape(-1, 3), _f32(directions).reshape(-1, 3), _f32(tmax).ravel(); n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        check(self.lib, self.lib.lumen_mi_query_any(self.h, n, _fp(o), _fp(d), tmin, _fp(tm), occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ

    def TestBsdf(self, mode, mat23, N, T, wo, aux):
        m, n_, t, w, a = _f32(mat23).reshape(-1, 23), _f32(N).reshape(-1, 3), _f32(T).reshape(-1, 3), _f32(wo).reshape(-1, 3), _f32(aux).reshape(-1, 3)
        out = np.zeros((m.shape[0], 8), np.float32)
        check(self.lib, self.lib.lumen_mi_test_bsdf(self.h, m.shape[0], mode, _fp(m), _fp(n_), _fp(t), _fp(w), _fp(a), _fp(out)))
        return out

    def TestRestir(self, mode, a, b=None, c=None):
        """Known-answer hook (lumen_mi_test_restir): 0 reservoir sequences (a, b, c: n x 8), 1 CDF queries (a: prefix sums, b: values), 2 sRGB8,
        3 / 5 Resample exact / fast (a: n x 35 surfaces, b: n x 14 samples -> n x 5), 4 / 6 CombineBiased of two reservoirs exact / fast
        (a: n x 35, b: n x 2 x 17, c: n seeds -> n x 18)."""
        a = _f32(a); n = a.shape[0] if mode in (0, 3, 4, 5, 6) else a.size
        b = None if b is None else _f32(b); m = 0 if (b is None or mode != 1) else b.size
        c = None if c is None else np.ascontiguousarray(c, np.uint32)
        nout = {0: 33 * n, 1: 2 * m, 2: n, 3: 5 * n, 5: 5 * n, 4: 18 * n, 6: 18 * n}[mode]
        out = np.zeros(nout, np.float32)
        check(self.lib, self.lib.lumen_mi_test_restir(self.h, mode, n, _fp(a), None if b is None else _fp(b),
                                                      None if c is None else c.ctypes.data_as(C.POINTER(C.c_uint32)), m, _fp(out)))
        return out.reshape(n, -1) if mode >= 3 else out

    def TestRestirFrame(self, W, H, surf_cur, surf_prev, motion, lights, cdf, seed, current_index, occluded0, occluded1, res4, fast=0):
        """Known-answer hook (lumen_mi_test_restir_frame): the kernels of one ReSTIR::Run on rows of 32-bit words; see include/lumen_mi.h.
        Returns a dict: res4 (the four buffers afterwards), bags, stages [5][n][17], rays [pass] -> [count][8], direct [n][4]."""
        u32 = lambda a: np.ascontiguousarray(a, np.uint32)
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        n = W * H
        surf_cur = u32(surf_cur); motion = u32(motion); lights = u32(lights); cdf = u32(cdf)
        surf_prev = None if surf_prev is None else u32(surf_prev)
        o0 = np.ascontiguousarray(occluded0, np.uint8); o1 = np.ascontiguousarray(occluded1, np.uint8)
        res4 = u32(res4).copy()
        out = {"res4": res4, "bags": np.zeros((50000, 2), np.uint32), "stages": np.zeros((5, n, 17), np.uint32), "direct": np.zeros((n, 4), np.uint32)}
        rays = np.zeros((2, n, 8), np.uint32); counts = np.zeros(2, np.uint32)
        check(self.lib, self.lib.lumen_mi_test_restir_frame(self.h, W, H, up(surf_cur), None if surf_prev is None else up(surf_prev), up(motion), lights.shape[0], up(lights), up(cdf),
                                                            int(seed), int(current_index), u8p(o0), u8p(o1), int(fast), up(res4), up(out["bags"]), up(out["stages"]), up(rays), up(counts),
                                                            up(out["direct"])))
        out["rays"] = [rays[p, :int(counts[p])] for p in (0, 1)]
        return out

    def TestShade(self, W, H, rows43, lights, cdf, fast=0, direct=True, indirect=True):
        """Known-answer hook (lumen_mi_test_shade): ShadeDirect / ShadeIndirect on rows (x, y, seed, surface(40)) -> ([n][12], [n][10])."""
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        rows43 = np.ascontiguousarray(rows43, np.uint32); lights = np.ascontiguousarray(lights, np.uint32); cdf = np.ascontiguousarray(cdf, np.uint32)
        n = rows43.shape[0]
        d = np.zeros((n, 12), np.uint32) if direct else None; i = np.zeros((n, 10), np.uint32) if indirect else None
        check(self.lib, self.lib.lumen_mi_test_shade(self.h, n, W, H, up(rows43), lights.shape[0], up(lights), up(cdf), int(fast), None if d is None else up(d), None if i is None else up(i)))
        return d, i

    def TestExtract(self, hits9, rays9):
        """Known-answer hook (lumen_mi_test_extract): lm_extract on (hit record, ray) rows against the current scene -> [n][35] words."""
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        hits9 = np.ascontiguousarray(hits9, np.uint32); rays9 = np.ascontiguousarray(rays9, np.uint32); out = np.zeros((hits9.shape[0], 35), np.uint32)
        check(self.lib, self.lib.lumen_mi_test_extract(self.h, hits9.shape[0], up(hits9), up(rays9), up(out)))
        return out

    def TestTex2D(self, texture, uv):
        """Known-answer hook (lumen_mi_test_tex2d): the extraction kernels' texture fetch on [n][2] normalised coordinates of one texture -> [n][4] float32."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2); out = np.zeros((uv.shape[0], 4), np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(self.lib, self.lib.lumen_mi_test_tex2d(self.h, texture, uv.shape[0], fp(uv), fp(out)))
        return out

    def TestExtract0(self, hits9, dirs3, eye3, matrix16):
        """Known-answer hook (lumen_mi_test_extract0): the depth-0 kernel on hit records for every pixel -> (gbuffer [n][8][4], motion half2 bits [n], direct [n][4])."""
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        hits9 = np.ascontiguousarray(hits9, np.uint32); dirs3 = np.ascontiguousarray(dirs3, np.uint32); n = hits9.shape[0]
        eye3 = np.ascontiguousarray(eye3, np.uint32); matrix16 = np.ascontiguousarray(matrix16, np.uint32)
        g = np.zeros((n, 8, 4), np.float32); mv = np.zeros(n, np.uint32); d = np.zeros((n, 4), np.float32)
        check(self.lib, self.lib.lumen_mi_test_extract0(self.h, up(hits9), up(dirs3), up(eye3), up(matrix16), _fp(g), up(mv), _fp(d)))
        return g, mv, d

    def TestPrimaryRays(self, W, H, frame_count, cam12):
        """Known-answer hook (lumen_mi_test_primary_rays): the primary-ray kernel on a W x H image -> [n][11] words (x y origin direction contribution)."""
        cam12 = np.ascontiguousarray(cam12, np.uint32); out = np.zeros((W * H, 11), np.uint32)
        check(self.lib, self.lib.lumen_mi_test_primary_rays(self.h, W, H, int(frame_count), cam12.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def TestMath(self, fn, x, y=None):
        x = _f32(x).ravel(); y = x if y is None else _f32(y).ravel(); out = np.zeros_like(x)
        check(self.lib, self.lib.lumen_mi_test_math(self.h, x.size, fn, _fp(x), _fp(y), _fp(out)))
        return out

    def GetWorldTriangles(self):
        n = C.c_uint32(); check(self.lib, self.lib.lumen_mi_get_world_triangles(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3, 3), np.float32)
        check(self.lib, self.lib.lumen_mi_get_world_triangles(self.h, _fp(out), n.value, C.byref(n)))
        return out

    def GetLights(self):
        n = C.c_uint32(); check(self.lib, self.lib.lumen_mi_get_lights(self.h, None, None, 0, C.byref(n)))
        lights, cdf = np.zeros((n.value, 16), np.float32), np.zeros(n.value, np.float32)
        check(self.lib, self.lib.lumen_mi_get_lights(self.h, _fp(lights), _fp(cdf), n.value, C.byref(n)))
        return lights, cdf

    def GetBvhInfo(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self.lib, self.lib.lumen_mi_get_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(nodes=a.value, triangles=b.value, max_depth=c.value)

    # ---- convenience: replay a SceneDescription through the factories, exactly as SceneManager would ------------------
    def LoadSceneDescription(self, desc):
        tex = [self.CreateTexture(t["pixels"], normalize=t["srgb"]) for t in desc.textures]
        mats = []
        for m in desc.materials:
            kw = dict(m)
            for f in ("diffuse_texture", "normal_map", "metallic_roughness_texture", "emissive_texture", "transmission_texture",
                      "clearcoat_texture", "clearcoat_roughness_texture", "tint_texture"):
                kw[f] = tex[m[f]]
            mats.append(self.CreateMaterial(**kw))
        prims = [self.CreatePrimitive(mats[p["material"]], p["indices"], vertices=p["vertices"], index_size=p.get("index_size", 4))[0] for p in desc.primitives]
        meshes = [self.CreateMesh([prims[i] for i in m]) for m in desc.meshes]
        scene = self.CreateScene()
        for inst in desc.instances:
            mi = scene.AddMesh(meshes[inst["mesh"]])
            mi.SetTransform(inst["transform"])
            if inst["override_material"] >= 0:
                mi.SetOverrideMaterial(mats[inst["override_material"]])
            mi.SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])
        self.SetScene(scene)
        self.m_Scene = scene
        self.m_Materials = mats
        self.m_Meshes = meshes
        cam = desc.camera
        self.SetCamera(cam["position"], cam["right"], cam["up"], cam["forward"], cam["fov"])
        return scene
