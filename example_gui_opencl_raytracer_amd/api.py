"""ctypes mirror of the C-ABI boundary (include/opencl_wrap.h, include/hip_wrap_ext.h).

The functions keep the reference's names, argument order and meaning
(reference src/opencl_wrap.h:29-42).  ``ClWrap`` is a thin object wrapper so tests read
like the reference drivers (raypng.c:31-89).  There is no CPU fallback here: if the
HIP library is missing the import of the library raises, and without a GPU
``cl_wrap_init`` terminates the process exactly like the reference does without an
OpenCL device (opencl_wrap.c:31-34).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLWRAP_LIB") or os.path.join(HERE, "libopencl_wrap_hip.so")   # CLWRAP_LIB: A/B builds

MAX_KERNELS = 16  # __MAX_KERNELS, opencl_wrap.h:6
MAX_BUFFERS = 32  # __MAX_BUFFERS, opencl_wrap.h:7

CL_DEVICE_TYPE_GPU = 1 << 2
CL_MEM_READ_WRITE = 1 << 0
CL_MEM_WRITE_ONLY = 1 << 1
CL_MEM_READ_ONLY = 1 << 2
CL_MEM_COPY_HOST_PTR = 1 << 5

# every symbol include/*.h declares (checked by the CPU test-suite)
SYMBOLS = [
    "cl_wrap_init", "cl_wrap_load_global_data", "cl_wrap_load_single_data", "cl_wrap_load_images",
    "cl_wrap_output", "cl_wrap_release",
    "clw_ext_set_depth", "clw_ext_get_depth", "clw_ext_set_strict", "clw_ext_set_fuse",
    "clw_ext_set_id_offset", "clw_ext_set_row_bands", "clw_ext_set_async", "clw_ext_sync", "clw_ext_set_stream",
    "clw_ext_timing_reset", "clw_ext_timing_get", "clw_ext_set_timing_every", "clw_ext_set_pipeline", "clw_ext_load_images_raw",
    "clw_ext_bind_device_buffer", "clw_ext_device_ptr", "clw_ext_set_debug_rgb",
    "clw_ext_enable_counters", "clw_ext_read_counters", "clw_ext_set_tile_sched", "clw_ext_read_tile_costs", "clw_ext_unit", "clw_ext_set_grid", "clw_ext_set_variant",
    "clw_ext_set_shadow_through", "clw_ext_set_tpt", "clw_ext_invalidate_scene", "clw_ext_read_counters_ex", "clw_ext_unit_scene",
    "clw_host_perspective", "clw_host_write_png", "clw_host_write_png_rgba", "clw_host_read_png",
    "clw_host_free", "clw_ext_version",
]


def library_version() -> str:
    """clw_ext_version(): 'opencl_wrap_hip <ver> gfx950 fast+strict kernels:<sha256/16 of the device sources>'."""
    return load_library().clw_ext_version().decode()


def kernel_source_hash() -> str:
    return library_version().rsplit("kernels:", 1)[-1]


class cl_wrap(C.Structure):
    """Layout of ``struct cl_wrap`` in include/opencl_wrap.h."""
    _fields_ = [
        ("impl", C.c_void_p),
        ("kernels_num", C.c_uint32),
        ("buffers_num", C.c_uint32 * MAX_KERNELS),
        ("buffers_ids", (C.c_uint32 * MAX_BUFFERS) * MAX_KERNELS),
        ("buffers", (C.c_void_p * MAX_BUFFERS) * MAX_KERNELS),
    ]


class clw_camera(C.Structure):
    _fields_ = [("im_corner", C.c_float * 3), ("origin", C.c_float * 3), ("up", C.c_float * 3),
                ("right", C.c_float * 3), ("w_factor", C.c_float), ("h_factor", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32)]


_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load the HIP shim.  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m example_gui_opencl_raytracer_amd.build` "
                          "(there is no CPU fallback for the trace path)")
    L = C.CDLL(path)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    W = C.POINTER(cl_wrap)
    L.cl_wrap_load_global_data.argtypes = [W, u32, u32, vp, sz, C.c_uint64]
    L.cl_wrap_load_single_data.argtypes = [W, u32, u32, vp, sz]
    L.cl_wrap_output.argtypes = [W, sz, sz, u32, u32, C.c_int32, vp]
    L.cl_wrap_release.argtypes = [W]
    L.clw_ext_set_depth.argtypes = [W, C.c_int]
    L.clw_ext_get_depth.argtypes = [W]
    L.clw_ext_get_depth.restype = C.c_int
    for name in ("clw_ext_set_strict", "clw_ext_set_fuse", "clw_ext_set_async", "clw_ext_enable_counters",
                 "clw_ext_set_variant", "clw_ext_set_tile_sched", "clw_ext_set_grid"):
        getattr(L, name).argtypes = [W, C.c_int]
    L.clw_ext_set_shadow_through.argtypes = [W, C.c_float]
    if hasattr(L, "clw_ext_set_tpt") or not os.environ.get("CLWRAP_LIB"):      # (an older A/B build may lack it)
        L.clw_ext_set_tpt.argtypes = [W, C.c_int, C.c_int, C.c_int]
    L.clw_ext_set_id_offset.argtypes = [W, C.c_uint64]
    L.clw_ext_set_row_bands.argtypes = [W, u32, u32]
    L.clw_ext_sync.argtypes = [W]
    L.clw_ext_set_stream.argtypes = [W, vp]
    L.clw_ext_timing_reset.argtypes = [W]
    L.clw_ext_set_timing_every.argtypes = [W, u32]
    L.clw_ext_set_pipeline.argtypes = [W, C.c_int]
    L.clw_ext_timing_get.argtypes = [W, u32, C.POINTER(u32), C.POINTER(C.c_double)]
    L.clw_ext_load_images_raw.argtypes = [W, u32, u32, vp, u32, u32, u32]
    L.clw_ext_bind_device_buffer.argtypes = [W, u32, u32, vp, sz]
    L.clw_ext_device_ptr.argtypes = [W, u32, u32]
    L.clw_ext_device_ptr.restype = vp
    L.clw_ext_set_debug_rgb.argtypes = [W, vp]
    L.clw_ext_unit.argtypes = [W, C.c_int, vp, u32, vp, u32, u32, u32]
    L.clw_ext_read_tile_costs.argtypes = [W, vp, u32]
    L.clw_ext_read_tile_costs.restype = u32
    L.clw_ext_read_counters.argtypes = [W, C.POINTER(C.c_uint64 * 8)]
    L.clw_ext_read_counters_ex.argtypes = [W, C.POINTER(C.c_uint64 * 32), u32]
    L.clw_ext_invalidate_scene.argtypes = [W]
    L.clw_ext_unit_scene.argtypes = [W, u32, C.c_int, vp, u32, vp, u32, u32]
    L.clw_host_perspective.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float, C.c_float, u32, u32,
                                       C.POINTER(clw_camera)]
    L.clw_host_perspective.restype = C.c_int
    L.clw_host_write_png.argtypes = [C.c_char_p, vp, u32, u32]
    L.clw_host_write_png.restype = C.c_int
    L.clw_host_write_png_rgba.argtypes = [C.c_char_p, vp, u32, u32]
    L.clw_host_write_png_rgba.restype = C.c_int
    L.clw_host_read_png.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(vp)]
    L.clw_host_read_png.restype = C.c_int
    L.clw_host_free.argtypes = [vp]
    L.clw_ext_version.restype = C.c_char_p
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.cast(a, C.c_void_p)


def perspective(origin, look, fov, focal, width, height) -> clw_camera:
    """rinit_camera + rgen_perspective (reference src/cpu_ray.c:24-35, 42-106)."""
    L = load_library()
    cam = clw_camera()
    ok = L.clw_host_perspective((C.c_float * 3)(*origin), (C.c_float * 3)(*look), fov, focal, width, height,
                                C.byref(cam))
    if not ok:
        raise ValueError("rgen_perspective rejects this camera (cpu_ray.c:58-63)")
    return cam


def write_png(path: str, xrgb: np.ndarray, width: int, height: int) -> None:
    xrgb = np.ascontiguousarray(xrgb, np.uint32)
    assert xrgb.size == width * height
    rc = load_library().clw_host_write_png(path.encode(), _ptr(xrgb), width, height)
    if rc:
        raise OSError(f"png write failed ({rc}): {path}")


def write_png_rgba(path: str, rgba: np.ndarray) -> None:
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    rc = load_library().clw_host_write_png_rgba(path.encode(), _ptr(rgba), w, h)
    if rc:
        raise OSError(f"png write failed ({rc}): {path}")


def read_png(path: str) -> np.ndarray:
    """-> uint8 [h, w, 4] (A = 255); raises ValueError with the decoder's code otherwise."""
    L = load_library()
    w, h, p = C.c_uint32(), C.c_uint32(), C.c_void_p()
    rc = L.clw_host_read_png(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc:
        raise ValueError(f"png read failed ({rc}): {path}")
    try:
        buf = (C.c_uint8 * (w.value * h.value * 4)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        L.clw_host_free(p)


class ClWrap:
    """Object form of the six cl_wrap_* calls plus the extensions."""

    def __init__(self, *sources_and_names, device_type=CL_DEVICE_TYPE_GPU):
        self.L = load_library()
        self.w = cl_wrap()
        if not sources_and_names:
            sources_and_names = ("src/cl/raygen.cl", "raygen", "src/cl/raytracing.cl", "raytracer")
        args = [C.c_char_p(s.encode()) for s in sources_and_names] + [C.c_char_p(None)]
        self.L.cl_wrap_init(C.byref(self.w), C.c_uint64(device_type), *args)
        self._keep = []

    # ---- the reference API ----
    def load_global_data(self, kernel_id, arg_id, data, size=None, mem_flags=CL_MEM_READ_WRITE):
        if isinstance(data, np.ndarray):
            data = np.ascontiguousarray(data)
            size = data.nbytes if size is None else size
        self.L.cl_wrap_load_global_data(C.byref(self.w), kernel_id, arg_id, _ptr(data), size, mem_flags)

    def load_single_data(self, kernel_id, arg_id, obj):
        """obj: a ctypes object / bytes / numpy scalar or array (passed by value)."""
        if isinstance(obj, (bytes, bytearray)):
            buf = C.create_string_buffer(bytes(obj), len(obj))
            self.L.cl_wrap_load_single_data(C.byref(self.w), kernel_id, arg_id, C.cast(buf, C.c_void_p), len(obj))
        elif isinstance(obj, np.ndarray) or isinstance(obj, np.generic):
            a = np.ascontiguousarray(obj)
            self.L.cl_wrap_load_single_data(C.byref(self.w), kernel_id, arg_id, _ptr(a), a.nbytes)
        else:
            self.L.cl_wrap_load_single_data(C.byref(self.w), kernel_id, arg_id, C.cast(C.byref(obj), C.c_void_p),
                                            C.sizeof(obj))

    def load_images(self, kernel_id, arg_id, *paths, mem_flags=CL_MEM_COPY_HOST_PTR):
        args = [C.c_char_p(p.encode()) for p in paths]
        self.L.cl_wrap_load_images(C.byref(self.w), C.c_uint32(kernel_id), C.c_uint32(arg_id),
                                   C.c_uint64(mem_flags), C.c_uint32(len(paths)), *args)

    def output(self, array_size, output_size, kernel_run_id, kernel_id, arg_id, host_output=None):
        self.L.cl_wrap_output(C.byref(self.w), array_size, output_size, kernel_run_id, kernel_id, arg_id,
                              _ptr(host_output))

    def release(self):
        if self.w.impl:
            self.L.cl_wrap_release(C.byref(self.w))

    def buffer_handle(self, kernel_id, arg_id) -> C.c_void_p:
        """&wrap.buffers[k][a] as the drivers use it (raypng.c:61)."""
        return C.c_void_p(self.w.buffers[kernel_id][arg_id])

    # ---- extensions ----
    def set_depth(self, d): self.L.clw_ext_set_depth(C.byref(self.w), d)
    def get_depth(self): return self.L.clw_ext_get_depth(C.byref(self.w))
    def set_strict(self, s): self.L.clw_ext_set_strict(C.byref(self.w), int(s))
    def set_fuse(self, f): self.L.clw_ext_set_fuse(C.byref(self.w), int(f))
    def set_id_offset(self, first_id): self.L.clw_ext_set_id_offset(C.byref(self.w), first_id)
    def set_row_bands(self, stride, phase): self.L.clw_ext_set_row_bands(C.byref(self.w), stride, phase)
    def set_async(self, a): self.L.clw_ext_set_async(C.byref(self.w), int(a))
    def sync(self): self.L.clw_ext_sync(C.byref(self.w))
    def set_stream(self, s): self.L.clw_ext_set_stream(C.byref(self.w), C.c_void_p(s))
    def unit(self, op: int, rows: np.ndarray, out_cols: int, aux: int = 0) -> np.ndarray:
        """Run one device helper (see clw_ext_unit) over float32 rows -> float32 [n, out_cols]."""
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.zeros((rows.shape[0], out_cols), np.float32)
        self.L.clw_ext_unit(C.byref(self.w), op, _ptr(rows), rows.shape[1], _ptr(out), out_cols, rows.shape[0], aux)
        return out

    def read_tile_costs(self) -> np.ndarray:
        n = self.L.clw_ext_read_tile_costs(C.byref(self.w), None, 0)
        out = np.zeros(n, np.uint32)
        if n:
            self.L.clw_ext_read_tile_costs(C.byref(self.w), _ptr(out), n)
        return out

    def set_shadow_through(self, f): self.L.clw_ext_set_shadow_through(C.byref(self.w), float(f))
    def set_grid(self, on): self.L.clw_ext_set_grid(C.byref(self.w), int(on))
    def set_tile_sched(self, on): self.L.clw_ext_set_tile_sched(C.byref(self.w), int(on))
    def set_variant(self, v): self.L.clw_ext_set_variant(C.byref(self.w), int(v))
    def set_tpt(self, max_lanes=-1, min_paths=-1, pool_mb=-1): self.L.clw_ext_set_tpt(C.byref(self.w), int(max_lanes), int(min_paths), int(pool_mb))
    def timing_reset(self): self.L.clw_ext_timing_reset(C.byref(self.w))
    def set_timing_every(self, n): self.L.clw_ext_set_timing_every(C.byref(self.w), int(n))
    def set_pipeline(self, on): self.L.clw_ext_set_pipeline(C.byref(self.w), int(on))

    def timing_get(self, kernel_id):
        n, ms = C.c_uint32(), C.c_double()
        self.L.clw_ext_timing_get(C.byref(self.w), kernel_id, C.byref(n), C.byref(ms))
        return n.value, ms.value

    def load_images_raw(self, kernel_id, arg_id, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        layers, h, w, c = rgba.shape
        assert c == 4
        self.L.clw_ext_load_images_raw(C.byref(self.w), kernel_id, arg_id, _ptr(rgba), w, h, layers)

    def bind_device_buffer(self, kernel_id, arg_id, device_ptr, size):
        self.L.clw_ext_bind_device_buffer(C.byref(self.w), kernel_id, arg_id, C.c_void_p(device_ptr), size)

    def device_ptr(self, kernel_id, arg_id): return self.L.clw_ext_device_ptr(C.byref(self.w), kernel_id, arg_id)
    def set_debug_rgb(self, ptr): self.L.clw_ext_set_debug_rgb(C.byref(self.w), C.c_void_p(ptr))
    def enable_counters(self, on): self.L.clw_ext_enable_counters(C.byref(self.w), int(on))

    def read_counters(self):
        """Work counters of the counting build (clw_ext_read_counters_ex); `shadow_rays` counts every shadow ray the
        reference would cast, `shadow_rays_traced` leaves out the ones elided on zero-coefficient surfaces."""
        out = (C.c_uint64 * 32)()
        self.L.clw_ext_read_counters_ex(C.byref(self.w), C.byref(out), 32)
        self.last_raw_counters = [int(x) for x in out]
        names = ["segments", "shadow_rays", "light_probes", "sky_fetches", "texel_fetches", "pushes",
                 "lane_iters", "wave_iters_x64", "shadow_rays_traced", "lights_classified"]
        d = dict(zip(names, [int(x) for x in out]))
        d["vis_mismatches"] = int(out[28])
        d["tpt_gave_up"], d["tpt_tiles"], d["tpt_nodes"] = int(out[29]), int(out[30]), int(out[31])   # tree-parallel tail (deep launches)
        d["tpt_batches"], d["tpt_max_batches"], d["tpt_max_nodes"], d["tpt_longest_us"] = int(out[16]), int(out[17]), int(out[18]), round(int(out[19]) * 0.01, 1)
        d["tpt_phase_max_us"] = [round(int(out[20 + k]) * 0.01, 1) for k in range(6)]
        d["tpt_phase_us"] = [round(int(out[10 + k]) * 0.01, 1) for k in range(6)]   # roots, expansion, order, RNG, shading, replay (summed over tiles)
        return d

    def invalidate_scene(self): self.L.clw_ext_invalidate_scene(C.byref(self.w))

    def unit_scene(self, op: int, rows: np.ndarray, out_cols: int, kernel_id: int = 1) -> np.ndarray:
        """Run one scene-dependent device helper (see clw_ext_unit_scene) over float32 rows -> float32 [n, out_cols]."""
        rows = np.ascontiguousarray(rows, np.float32)
        out = np.zeros((rows.shape[0], out_cols), np.float32)
        self.L.clw_ext_unit_scene(C.byref(self.w), kernel_id, op, _ptr(rows), rows.shape[1], _ptr(out), out_cols, rows.shape[0])
        return out
