"""ctypes bindings for the checker libraries -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
  * ``Oracle``    -> oracle/liboracle.so   (the plain-C restatement, whitted_oracle.c)
  * ``Reference`` -> oracle/_ref/libref_cl.so (the reference's own .cl compiled for the host;
                     present only where it was built, see oracle/Makefile)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_cl.so")
REF_FMA_SO = os.path.join(HERE, "_ref", "libref_cl_fma.so")      # the same kernels built with FMA contraction
REF_CAMERA_SO = os.path.join(HERE, "_ref", "libref_cpu_ray.so")  # the reference's src/cpu_ray.c (host camera code)
MARGINS_SO = os.path.join(HERE, "liboracle_margins.so")         # the restatement with its decision-margin tracker

c_f = C.c_float
c_fp = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)


def build(ref: bool = True) -> None:
    """Compile the checker (gcc; and, where /root/reference exists, the _ref build)."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref:   # both are no-ops where /root/reference does not exist (the GPU box keeps the prebuilt binaries)
        subprocess.run(["make", "-s", "-C", HERE, "ref", "raypng", "rayinteractive"], check=True)


class Camera(C.Structure):
    _fields_ = [("im_corner", c_f * 3), ("origin", c_f * 3), ("up", c_f * 3), ("right", c_f * 3),
                ("w_factor", c_f), ("h_factor", c_f), ("width", C.c_uint32), ("height", C.c_uint32)]


class SceneC(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("ns", C.c_uint32), ("planes", C.c_void_p), ("np", C.c_uint32),
                ("lights", C.c_void_p), ("nl", C.c_uint32), ("tex", C.c_void_p), ("tex_w", C.c_int32),
                ("tex_h", C.c_int32), ("tex_layers", C.c_int32), ("sky", C.c_void_p),
                ("sky_w", C.c_int32), ("sky_h", C.c_int32)]


class Counters(C.Structure):
    _names = ["segments", "light_probes", "shadow_rays", "sky_fetches", "texel_fetches", "sphere_tests",
              "plane_tests", "shaded_hits", "pushes", "tir_drops", "int_cast_oor", "oob_reads", "max_stack"]
    _fields_ = [(n, C.c_uint64) for n in _names]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in self._names}

    @property
    def rays(self):
        """rays = path segments + shadow rays."""
        return int(self.segments + self.shadow_rays)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (c_f * 3)(*[float(x) for x in v])


class _Inputs:
    """Keeps numpy buffers alive and exposes them as a wo_scene."""

    def __init__(self, scene, tex: np.ndarray, sky: np.ndarray):
        self.spheres, self.planes, self.lights = scene.spheres, scene.planes, scene.lights
        self.tex = np.ascontiguousarray(tex, np.uint8)
        self.sky = np.ascontiguousarray(sky, np.uint8)
        assert self.tex.ndim == 4 and self.tex.shape[3] == 4
        assert self.sky.ndim == 4 and self.sky.shape[0] == 1 and self.sky.shape[3] == 4
        self.c = SceneC(_p(self.spheres).value, len(self.spheres), _p(self.planes).value, len(self.planes),
                        _p(self.lights).value, len(self.lights), _p(self.tex).value, self.tex.shape[2],
                        self.tex.shape[1], self.tex.shape[0], _p(self.sky).value, self.sky.shape[2],
                        self.sky.shape[1])


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.wo_perspective.restype = C.c_int
        L.wo_perspective.argtypes = [c_f * 3, c_f * 3, c_f, c_f, C.c_uint32, C.c_uint32, C.POINTER(Camera)]
        L.wo_normalize3.argtypes = [c_f * 3, c_f * 3]
        L.wo_render.restype = C.c_int
        L.wo_render.argtypes = [C.POINTER(Camera), C.POINTER(SceneC), C.c_int, C.c_uint64, C.c_uint64,
                                C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.wo_trace_rays.restype = C.c_int
        L.wo_trace_rays.argtypes = [C.c_void_p, C.POINTER(SceneC), C.c_int, C.c_uint64, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.wo_raygen.argtypes = [C.POINTER(Camera), C.c_uint64, C.c_uint64, C.c_void_p]
        L.wo_num_threads.restype = C.c_int
        for fn in ("wo_libm_sinf", "wo_libm_cosf"):
            getattr(L, fn).restype = c_f
            getattr(L, fn).argtypes = [c_f]
        L.wo_libm_powf.restype = c_f
        L.wo_libm_powf.argtypes = [c_f, c_f]
        L.wo_libm_angle.restype = c_f
        L.wo_libm_angle.argtypes = [c_f, C.c_int]
        L.wo_trace_libm.restype = C.c_int
        L.wo_trace_libm.argtypes = [C.POINTER(Camera), C.POINTER(SceneC), C.c_int, C.c_uint64, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_uint32]
        L.wo_set_transparent_through.argtypes = [c_f]
        L.wo_intersect_sphere.restype = C.c_int
        L.wo_intersect_sphere.argtypes = [c_f * 3, c_f * 3, c_f * 3, c_f, c_fp]
        L.wo_intersect_plane.restype = C.c_int
        L.wo_intersect_plane.argtypes = [c_f * 3, c_f * 3, c_f * 3, c_f * 3, c_fp]
        L.wo_reflect.argtypes = [c_f * 3, c_f * 3, c_f * 3]
        L.wo_refract.argtypes = [c_f, c_f, c_f * 3, c_f * 3, c_f * 3]
        L.wo_schlick.restype = c_f
        L.wo_schlick.argtypes = [c_f, c_f, c_f * 3, c_f * 3]
        L.wo_map_to_cube.argtypes = [c_f * 3, C.c_uint32, C.c_int32 * 2]
        L.wo_xorshift32.restype = c_f
        L.wo_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
        L.wo_euclidean_modulo.restype = C.c_int
        L.wo_euclidean_modulo.argtypes = [C.c_int, C.c_int]
        L.wo_plane_texture_pixel.argtypes = [C.c_void_p, c_f * 3, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f * 3]
        L.wo_shadow.restype = c_f
        L.wo_shadow.argtypes = [c_f * 3, c_f * 3, C.POINTER(SceneC)]
        L.wo_find_light.restype = C.c_int
        L.wo_find_light.argtypes = [c_f * 3, c_f * 3, C.POINTER(SceneC), c_f * 3]
        L.wo_find_solid.restype = C.c_int
        L.wo_find_solid.argtypes = [c_f * 3, c_f * 3, C.POINTER(SceneC), c_f * 3, c_f * 3, C.c_void_p]

    # ---- camera (cpu_ray.c:24-35 + :42-106)
    def camera(self, origin, look, fov, focal, width, height, normalize=True) -> Camera:
        d = _f3(look)
        if normalize:
            dn = (c_f * 3)()
            self.lib.wo_normalize3(d, dn)
            d = dn
        cam = Camera()
        ok = self.lib.wo_perspective(_f3(origin), d, fov, focal, width, height, C.byref(cam))
        if not ok:
            raise ValueError("rgen_perspective rejects this camera (cpu_ray.c:58-63)")
        return cam

    def set_transparent_through(self, t: float) -> None:
        """Factor a transparent sphere applies to a shadow ray (0.8 = primitives.cl:7); a process-wide setting of this library."""
        self.lib.wo_set_transparent_through(float(t))

    def num_threads(self) -> int:
        return int(self.lib.wo_num_threads())

    def render(self, cam: Camera, scene, tex, sky, depth, id_begin=0, id_end=None, want_rgb=False,
               threads=0):
        """-> (uint32[n] packed 0x00RRGGBB, float32[n,3] or None, Counters)"""
        inp = _Inputs(scene, tex, sky)
        total = cam.width * cam.height
        id_end = total if id_end is None else id_end
        n = id_end - id_begin
        out = np.zeros(n, np.uint32)
        rgb = np.zeros((n, 3), np.float32) if want_rgb else None
        cnt = Counters()
        rc = self.lib.wo_render(C.byref(cam), C.byref(inp.c), depth, id_begin, id_end, _p(out),
                                _p(rgb) if want_rgb else None, C.byref(cnt), threads)
        if rc:
            raise ValueError("wo_render: bad depth / id range")
        return out, rgb, cnt

    def trace_libm(self, cam: Camera, scene, tex, sky, depth, pixel_id, cap=1 << 16, overrides=None):
        """-> (packed pixel, float32 [rows, 4] = {tag, a, b, result}): the sinf / cosf / powf calls of ONE pixel and the results used
        (tag 0: the xorshift pair of a soft-shadow sample; 1-4: sin / cos of phi and theta; 5: powf(a, b)).  `overrides`: rows of the
        same shape whose `result` replaces glibc's for exactly these input bits (another libm's values)."""
        inp = _Inputs(scene, tex, sky)
        log = np.zeros((cap, 4), np.float32)
        px = C.c_uint32(0)
        ov = None if overrides is None or len(overrides) == 0 else np.ascontiguousarray(overrides, np.float32)
        n = self.lib.wo_trace_libm(C.byref(cam), C.byref(inp.c), depth, int(pixel_id), C.byref(px), _p(log), cap,
                                   _p(ov) if ov is not None else None, 0 if ov is None else len(ov))
        if n < 0 or n > cap:
            raise ValueError("wo_trace_libm: bad depth, or the pixel makes more libm calls than `cap`")
        return int(px.value), log[:n].copy()

    def raygen(self, cam: Camera, id_begin=0, id_end=None) -> np.ndarray:
        id_end = cam.width * cam.height if id_end is None else id_end
        rays = np.zeros((id_end - id_begin, 16), np.float32)
        self.lib.wo_raygen(C.byref(cam), id_begin, id_end, _p(rays))
        return rays

    def trace_rays(self, rays16: np.ndarray, scene, tex, sky, depth, id_begin=0, threads=0):
        inp = _Inputs(scene, tex, sky)
        rays16 = np.ascontiguousarray(rays16, np.float32)
        n = rays16.shape[0]
        out = np.zeros(n, np.uint32)
        cnt = Counters()
        rc = self.lib.wo_trace_rays(_p(rays16), C.byref(inp.c), depth, id_begin, id_begin + n, _p(out), None,
                                    C.byref(cnt), threads)
        if rc:
            raise ValueError("wo_trace_rays: bad depth / id range")
        return out, cnt


def shifted_camera(cam: Camera, dx: float, dy: float) -> Camera:
    """The camera whose pixel (x, y) samples where `cam` samples (x + dx, y + dy): a sub-pixel shift of the image
    corner along the camera's own right / up steps (raygen.cl:13-17).  Used for the jitter masks of tests/golden."""
    c = Camera.from_buffer_copy(bytes(cam))
    for k in range(3):
        c.im_corner[k] = np.float32(np.float64(cam.im_corner[k]) + np.float64(cam.right[k]) * np.float64(cam.w_factor) * dx
                                    - np.float64(cam.up[k]) * np.float64(cam.h_factor) * dy)
    return c


MARGIN_SITES = ["disc", "root", "plane", "nearest", "shadow_t", "cast", "face", "tir"]   # WO_M_* of whitted_oracle.c


def render_margins(cam: Camera, scene, tex, sky, depth, through=None):
    """-> (uint32[n] frame, float32[n, 8] smallest relative decision margin per site class); diagnostic build."""
    L = C.CDLL(MARGINS_SO)
    L.wo_set_transparent_through.argtypes = [c_f]
    L.wo_set_transparent_through(0.8 if through is None else float(through))
    L.wo_render_margins.restype = C.c_int
    L.wo_render_margins.argtypes = [C.POINTER(Camera), C.POINTER(SceneC), C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    inp = _Inputs(scene, tex, sky)
    n = cam.width * cam.height
    out = np.zeros(n, np.uint32)
    mar = np.ones((n, len(MARGIN_SITES)), np.float32)
    if L.wo_render_margins(C.byref(cam), C.byref(inp.c), depth, 0, n, _p(out), _p(mar)):
        raise ValueError("wo_render_margins: bad depth")
    return out, mar


class RefCamera:
    """rinit_camera + rgen_perspective of the reference's own src/cpu_ray.c (built by oracle/Makefile)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_CAMERA_SO)

    def __init__(self, path: str = REF_CAMERA_SO):
        self.lib = C.CDLL(path)
        self.lib.ref_perspective.argtypes = [c_f * 3, c_f * 3, c_f, c_f, C.c_uint, C.c_uint, c_f * 14]

    def perspective(self, origin, look, fov, focal, width, height) -> np.ndarray:
        """-> float32[14]: im_corner, origin, up, right, w_factor, h_factor"""
        out = (c_f * 14)()
        self.lib.ref_perspective(_f3(origin), _f3(look), fov, focal, width, height, out)
        return np.array(out[:], np.float32)


class Reference:
    """The reference's own kernels compiled for the host (depths 1,2,3,4,8,15 only)."""

    DEPTHS = (1, 2, 3, 4, 8, 15)

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def __init__(self, path: str = REF_SO):
        self.lib = L = C.CDLL(path)
        L.ref_render.restype = C.c_int
        L.ref_render.argtypes = [C.c_int, c_f * 3, c_f * 3, c_f * 3, c_f * 3, c_f, c_f, C.c_uint, C.c_uint,
                                 C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint,
                                 C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_ulong)]
        L.ref_raygen.argtypes = [c_f * 3, c_f * 3, c_f * 3, c_f * 3, c_f, c_f, C.c_uint, C.c_uint, C.c_size_t,
                                 C.c_size_t, C.c_void_p]
        L.ref_intersect_sphere.restype = C.c_int
        L.ref_intersect_sphere.argtypes = [c_f * 3, c_f * 3, c_f * 3, c_f, c_fp]
        L.ref_intersect_plane.restype = C.c_int
        L.ref_intersect_plane.argtypes = [c_f * 3, c_f * 3, c_f * 3, c_f * 3, c_fp]
        L.ref_reflect.argtypes = [c_f * 3, c_f * 3, c_f * 3]
        L.ref_refract.argtypes = [c_f, c_f, c_f * 3, c_f * 3, c_f * 3]
        L.ref_schlick.restype = c_f
        L.ref_schlick.argtypes = [c_f, c_f, c_f * 3, c_f * 3]
        L.ref_map_to_cube.argtypes = [c_f * 3, C.c_uint, C.c_int * 2]
        L.ref_xorshift32.restype = c_f
        L.ref_xorshift32.argtypes = [C.POINTER(C.c_uint)]
        L.ref_euclidean_modulo.restype = C.c_int
        L.ref_euclidean_modulo.argtypes = [C.c_int, C.c_int]
        L.ref_plane_texture_pixel.argtypes = [C.c_void_p, c_f * 3, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f * 3]
        L.ref_shadow.restype = c_f
        L.ref_shadow.argtypes = [c_f * 3, c_f * 3, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
        L.ref_find_light.restype = C.c_int
        L.ref_find_light.argtypes = [c_f * 3, c_f * 3, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                     C.c_uint, c_f * 3]
        L.ref_find_solid.restype = C.c_int
        L.ref_find_solid.argtypes = [c_f * 3, c_f * 3, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, c_f * 3, c_f * 3, C.c_void_p]

    def render(self, cam: Camera, scene, tex, sky, depth, id_begin=0, id_end=None):
        inp = _Inputs(scene, tex, sky)
        total = cam.width * cam.height
        id_end = total if id_end is None else id_end
        out = np.zeros(id_end - id_begin, np.uint32)
        oob = C.c_ulong(0)
        rc = self.lib.ref_render(depth, cam.im_corner, cam.origin, cam.up, cam.right, cam.w_factor,
                                 cam.h_factor, cam.width, cam.height, _p(inp.spheres), len(inp.spheres),
                                 _p(inp.planes), len(inp.planes), _p(inp.lights), len(inp.lights),
                                 _p(inp.tex), inp.tex.shape[2], inp.tex.shape[1], inp.tex.shape[0],
                                 _p(inp.sky), inp.sky.shape[2], inp.sky.shape[1], id_begin, id_end, _p(out),
                                 C.byref(oob))
        if rc:
            raise ValueError(f"reference build has no depth-{depth} instantiation")
        return out, int(oob.value)

    def raygen(self, cam: Camera, id_begin=0, id_end=None) -> np.ndarray:
        id_end = cam.width * cam.height if id_end is None else id_end
        rays = np.zeros((id_end - id_begin, 16), np.float32)
        self.lib.ref_raygen(cam.im_corner, cam.origin, cam.up, cam.right, cam.w_factor, cam.h_factor,
                            cam.width, cam.height, id_begin, id_end, _p(rays))
        return rays
