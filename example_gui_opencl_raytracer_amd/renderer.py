"""The raypng.c call protocol as a reusable object.

Every device interaction goes through the six ``cl_wrap_*`` entry points in the order the
reference driver uses them (reference raypng.c:31-89): kernel 0 ("raygen") arguments 0-7
by value and 8 = ray buffer; kernel 1 ("raytracer") argument 0 = the 8-byte handle of
``buffers[0][8]``, 1-3 scene arrays, 4-6 counts, 7 pixel count, 8 texture array,
9 skybox, 10 framebuffer; then ``output(run 0)`` and ``output(run 1, read [1][10])``.

Row strips (multi-GPU): a renderer may own rows ``[first_row, first_row + rows)`` of the
frame; work-item ids, and with them ``id % W``, ``id / W`` and the RNG seed, stay global
(SURVEY.md 8(e)), so a strip is bit-identical to the same rows of a full-frame render.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api
from .scene import RAY, Scene


def strip_rows(height: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row strip of `rank` out of `world`: (first_row, rows).  Rows are dealt in
    multiples of 8 (the tile height) while they last, so every strip but the last is tile-aligned."""
    tiles = (height + 7) // 8
    base, extra = divmod(tiles, world)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    r0, r1 = min(t0 * 8, height), min(t1 * 8, height)
    return r0, r1 - r0


class Renderer:
    def __init__(self, scene: Scene, tex: np.ndarray, sky: np.ndarray, width: int, height: int, *,
                 depth: int = 15, strict: bool = False, fuse: bool = True, first_row: int = 0,
                 rows: int | None = None, bands: tuple[int, int] | None = None, framebuffer_ptr: int | None = None, wide_counts: bool | None = None,
                 texture_paths=None, skybox_path=None):
        self.width, self.height = width, height
        self.first_row = first_row
        self.rows = height - first_row if rows is None else rows
        if bands is not None:                              # (stride, phase): every stride-th 8-row band
            assert first_row == 0 and rows is None and height % (8 * bands[0]) == 0
            self.rows = height // bands[0]
        self.pixels = self.rows * width                   # work-items of this renderer
        self.scene = scene
        w = self.w = api.ClWrap("src/cl/raygen.cl", "raygen", "src/cl/raytracing.cl", "raytracer")
        w.set_depth(depth)
        w.set_strict(strict)
        w.set_fuse(fuse)
        w.set_id_offset(first_row * width)
        if bands is not None:
            w.set_row_bands(*bands)

        u32 = lambda v: np.uint32(v)
        w.load_single_data(0, 6, u32(width))
        w.load_single_data(0, 7, u32(height))
        w.load_global_data(0, 8, None, RAY.itemsize * self.pixels, api.CL_MEM_READ_WRITE)
        w.load_single_data(1, 0, w.buffer_handle(0, 8))
        ns, np_, nl = scene.counts
        w.load_global_data(1, 1, scene.spheres, mem_flags=api.CL_MEM_READ_ONLY)
        w.load_global_data(1, 2, scene.planes, mem_flags=api.CL_MEM_READ_ONLY)
        w.load_global_data(1, 3, scene.lights, mem_flags=api.CL_MEM_READ_ONLY)
        wide = (max(ns, np_, nl) > 255) if wide_counts is None else wide_counts
        cnt = (lambda v: np.uint32(v)) if wide else (lambda v: np.uint8(v))   # uchar in the reference
        w.load_single_data(1, 4, cnt(ns))
        w.load_single_data(1, 5, cnt(np_))
        w.load_single_data(1, 6, cnt(nl))
        # `pixels` = the guard `id >= total_size` (raytracing.cl:24); work-items here are strip-local
        w.load_single_data(1, 7, u32(self.pixels))
        if texture_paths is not None:
            w.load_images(1, 8, *texture_paths)
        else:
            w.load_images_raw(1, 8, tex)
        if skybox_path is not None:
            w.load_images(1, 9, skybox_path)
        else:
            w.load_images_raw(1, 9, sky)
        if framebuffer_ptr is not None:
            w.bind_device_buffer(1, 10, framebuffer_ptr, 4 * self.pixels)
        else:
            w.load_global_data(1, 10, None, 4 * self.pixels, api.CL_MEM_WRITE_ONLY)
        self._rgb_dev = None

    # ---- camera: the six values of rgen_perspective, re-settable at any time (rayinteractive.c:98-103)
    def set_camera(self, cam) -> None:
        w = self.w
        f3 = lambda v: np.array([v[0], v[1], v[2], 0.0], np.float32)     # cl_float3 = 16 bytes
        w.load_single_data(0, 0, f3(cam.im_corner))
        w.load_single_data(0, 1, f3(cam.origin))
        w.load_single_data(0, 2, f3(cam.up))
        w.load_single_data(0, 3, f3(cam.right))
        w.load_single_data(0, 4, np.float32(cam.w_factor))
        w.load_single_data(0, 5, np.float32(cam.h_factor))

    def look(self, origin, look, fov=90.0, focal=1.0):
        cam = api.perspective(origin, look, fov, focal, self.width, self.height)
        self.set_camera(cam)
        return cam

    # ---- one frame: raygen launch + trace launch (+ blocking read-back)
    def render(self, readback: bool = True):
        self.w.output(self.pixels, 0, 0, 0, 0, None)
        if not readback:
            self.w.output(self.pixels, 0, 1, 1, 10, None)
            return None
        out = np.empty(self.pixels, np.uint32)
        self.w.output(self.pixels, out.nbytes, 1, 1, 10, out)
        return out

    def render_rgb(self):
        """-> (packed uint32[n], float32[n,3] un-clamped radiance): the optional float debug output."""
        if self._rgb_dev is None:
            self.w.load_global_data(1, 31, None, 12 * self.pixels, api.CL_MEM_WRITE_ONLY)  # spare arg slot
            self._rgb_dev = self.w.device_ptr(1, 31)
        self.w.set_debug_rgb(self._rgb_dev)
        out = self.render()
        rgb = np.empty((self.pixels, 3), np.float32)
        self.w.output(self.pixels, rgb.nbytes, 1, 1, 31, rgb)
        self.w.set_debug_rgb(0)
        return out, rgb

    def read_rays(self) -> np.ndarray:
        """The 64-B rray records of buffers[0][8] (materialised on demand in fused mode)."""
        rays = np.empty((self.pixels, 16), np.float32)
        self.w.output(self.pixels, rays.nbytes, 0, 0, 8, rays)
        return rays

    def release(self):
        self.w.release()
