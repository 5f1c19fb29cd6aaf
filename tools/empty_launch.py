"""Launch floor: 1920x1080 with an EMPTY scene (every primary ray misses -> one skybox fetch per pixel) -- what the
32 640 one-wave workgroups cost before any tracing (dispatch, kernarg loads, prologue, pack + store)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer

full = scene.render_map_scene()
empty = scene.Scene(full.spheres[:0], full.planes[:0], full.lights[:0])
tex, sky = textures.texture_layers(), textures.skybox_cross(4096)
for name, sc in (("empty", empty), ("render.map", full)):
    r = Renderer(sc, tex, sky, 1920, 1080, depth=4)
    r.look(**pkg.CAMERA_RAYPNG)
    for _ in range(5):
        r.render(readback=False)
    r.w.timing_reset(); r.w.set_async(1)
    t = time.perf_counter()
    for _ in range(200):
        r.render(readback=False)
    r.w.sync()
    wall = (time.perf_counter() - t) / 200
    n, ms = r.w.timing_get(1)
    print(json.dumps(dict(scene=name, kernel_ms=round(ms / n, 4), wall_ms=round(wall * 1e3, 4))), flush=True)
    r.release()
